#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration per load width (tools/fetch_calib.hip): usage tools/fetch_calib.sh <out.json>   (GPU box)
out=$1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/calib_f /tmp/calib_w
rocprofv3 --pmc FETCH_SIZE -d /tmp/calib_f -o p --output-format csv -- ./tools/fetch_calib > /tmp/calib_f.log 2>&1 || { tail -5 /tmp/calib_f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d /tmp/calib_w -o p --output-format csv -- ./tools/fetch_calib > /tmp/calib_w.log 2>&1 || { tail -5 /tmp/calib_w.log; exit 1; }
python3 - "$out" <<'P'
import csv, glob, json, sys
res = {}
for tag, d in (("FETCH_SIZE", "/tmp/calib_f"), ("WRITE_SIZE", "/tmp/calib_w")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != tag or "k_stream_read" not in r["Kernel_Name"]:
            continue
        nm = r["Kernel_Name"].replace(" ", "")
        width = 16 if "float,4u" in nm or "float4" in nm else (8 if "float,2u" in nm or "float2" in nm else 4)
        res.setdefault(str(width), {})[tag + "_KB"] = float(r["Counter_Value"])
known_r, known_w = float(1 << 30), 4.0 * 65536 * 256
for w, d in res.items():
    d["read_bytes"] = known_r
    d["write_bytes"] = known_w
    d["fetch_bytes_per_counter_byte"] = known_r / (d["FETCH_SIZE_KB"] * 1024.0)     # multiply FETCH_SIZE by this
    d["write_bytes_per_counter_byte"] = known_w / (d["WRITE_SIZE_KB"] * 1024.0)
json.dump({"_what": "known streamed bytes / rocprofv3 counter (KB x 1024) per bytes-per-lane of the loads; tools/fetch_calib.hip", "bytes_per_lane": res},
          open(sys.argv[1], "w"), indent=1)
print(json.dumps(res, indent=1))
P
