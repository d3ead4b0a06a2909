"""Shared builders for the parity tests: the same problem as an oracle instance (CPU restatement,
test infrastructure) and as a product instance (HIP path through the C ABI)."""
import numpy as np
import torch

from conftest import admm_info_from
from oracle import admm_oracle as orc


def rel(a, b):
    a = np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu().numpy() if torch.is_tensor(b) else b, dtype=np.float64)
    n = np.linalg.norm(b)
    d = np.linalg.norm(a - b)
    return d / n if n > 0 else d


def make_oracle(meta, mode, ablation="None", bug_compat=True, prefix="knn", T=None, t_in=None):
    info = admm_info_from(meta)
    T = int(meta["T"]) if T is None else T
    t_in = int(meta["t_in"]) if t_in is None else t_in
    if mode == "knn":
        return orc.OracleADMM(meta["knn_cl"], meta["knn_u_ew"], meta["knn_d_ew"], info, mode="knn", ablation=ablation,
                              t_in=t_in, T=T, bug_compat=bug_compat)
    if mode == "physical":
        return orc.OracleADMM(meta["phys_cl"], meta["phys_u_ew"], meta["phys_d_ew"], info, mode="physical",
                              ablation=ablation, t_in=t_in, T=T, bug_compat=bug_compat)
    return orc.OracleADMM(meta["knn_cl"], meta["knn_u_ew"], None, info, mode="line", ablation=ablation, t_in=t_in, T=T,
                          skip_connection=1 if mode == "line" else 3)


def make_product(meta, mode, ablation="None", compute_dtype=torch.float32, bug_compat=True, T=None, t_in=None, **kw):
    """mgadmm.ADMM_algorithm on the golden tables (tables injected so both sides see identical weights)."""
    from mgadmm.ADMM import ADMM_algorithm
    info = admm_info_from(meta)
    T = int(meta["T"]) if T is None else T
    t_in = int(meta["t_in"]) if t_in is None else t_in
    n = int(meta["n"])
    gi = {"n_nodes": n}
    if mode == "physical":
        cl, dl = torch.from_numpy(meta["phys_cl"]), torch.zeros(meta["phys_cl"].shape)
        u_ew, d_ew = meta["phys_u_ew"], meta["phys_d_ew"]
    else:
        cl, dl = torch.from_numpy(meta["knn_cl"]), torch.zeros(meta["knn_cl"].shape)
        u_ew, d_ew = meta["knn_u_ew"], meta.get("knn_d_ew") if hasattr(meta, "get") else meta["knn_d_ew"]
    blk = ADMM_algorithm(gi, info, use_kNN=(mode != "physical"), k=cl.shape[1] - 1, u_sigma=1.0, d_sigma=1.0,
                         ablation=ablation, t_in=t_in, T=T, use_line_graph=mode in ("line", "skip3"),
                         skip_connection=3 if mode == "skip3" else 1, tables=(cl, dl + 1.0),
                         compute_dtype=compute_dtype, bug_compat=bug_compat, **kw)
    # overwrite the weight tables with the golden ones (float32, exactly what the reference used)
    blk.u_ew = torch.from_numpy(np.asarray(u_ew, dtype=np.float32)).unsqueeze(0).repeat(T, 1, 1)
    if mode in ("knn", "physical"):
        blk.d_ew = torch.from_numpy(np.asarray(d_ew, dtype=np.float32)).unsqueeze(0).repeat(T - 1, 1, 1)
    return blk


def meta_from_g2(g):
    """g2/g3 fixture -> dict with the keys make_oracle/make_product expect."""
    d = {k: g[k] for k in ("rho", "rho_u", "rho_d", "mu_u", "mu_d1", "mu_d2", "T", "t_in", "n")}
    d["knn_cl"] = d["phys_cl"] = g["cl"]
    d["knn_u_ew"] = d["phys_u_ew"] = g["u_ew"]
    if g["d_ew"].ndim == 2:
        d["knn_d_ew"] = d["phys_d_ew"] = g["d_ew"]
    else:
        d["knn_d_ew"] = d["phys_d_ew"] = None
    return d


def case_inputs(meta, task, np_dtype):
    t_in = int(meta["t_in"])
    if task == "pred":
        return meta["x_true"][:, :t_in].astype(np_dtype), None
    mask = meta["mask"]
    return (meta["x_true"] * mask).astype(np_dtype), mask
