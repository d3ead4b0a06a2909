"""TrafficDataset (utils.py:54-134) with the series resident on the GPU.

Same constructor, attributes (`graph_info`, `data`, `transform`, `data_mean/std/max/min`) and methods
(`recover_data`, `get_predict_data`, `get_interpolated_data`) as the reference class; the per-node statistics,
the standardize / normalize transform and -- new, because the reference is B = 1 only -- whole batches of
sliding windows (`get_predict_batch`, `get_interpolated_batch`) are computed by HIP kernels behind the C ABI
(`mgadmm_series_stats`, `mgadmm_series_affine`, `mgadmm_gather_windows`, csrc/frontend.hip).  File parsing is
host work exactly as in the reference (pandas / numpy)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from . import utils as _u

__all__ = ["TrafficDataset", "series_stats", "gather_windows"]

_MG = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check_series(series):
    if not (series.is_cuda and series.is_contiguous() and series.dtype in _MG and series.dim() >= 2):
        raise ValueError("series must be a contiguous float32/float64 CUDA tensor (T, N[, C])")
    return series.shape[0], int(np.prod(series.shape[1:]))


def series_stats(series, std=True):
    """Per-column min, max, mean (and unbiased std) over time of a (T, N[, C]) device series.
    Returns tensors of shape (1, N[, C]) like `data.max(0, keepdim=True)[0]` etc. (utils.py:83-88)."""
    n_steps, n_cols = _check_series(series)
    shp = (1,) + tuple(series.shape[1:])
    outs = [torch.empty(shp, dtype=series.dtype, device=series.device) for _ in range(4 if std else 3)]
    _lib.check(_lib.lib.mgadmm_series_stats(_ptr(series), n_steps, n_cols, _MG[series.dtype], _ptr(outs[0]), _ptr(outs[1]),
                                            _ptr(outs[2]), _ptr(outs[3]) if std else None, _stream(series.device)))
    return tuple(outs)


def gather_windows(series, starts, win, mask=None):
    """out[b] = series[starts[b] : starts[b] + win] (* mask) for a batch of start indices -> (B, win, N[, C])."""
    n_steps, n_cols = _check_series(series)
    st = torch.as_tensor(starts, dtype=torch.int64).reshape(-1).to(series.device).contiguous()
    if st.numel() == 0:
        raise ValueError("no start indices")
    out = torch.empty((st.numel(), win) + tuple(series.shape[1:]), dtype=series.dtype, device=series.device)
    m = None
    if mask is not None:
        m = mask.to(device=series.device, dtype=torch.float32).contiguous()
        if m.numel() != win * n_cols:
            raise ValueError(f"mask has {m.numel()} elements, a window has {win * n_cols}")
    _lib.check(_lib.lib.mgadmm_gather_windows(_ptr(series), n_steps, n_cols, _MG[series.dtype], _ptr(st), st.numel(), int(win),
                                              _ptr(m), _ptr(out), _stream(series.device)))
    return out


class TrafficDataset:
    """Drop-in for the reference's TrafficDataset with `device=` (default: current CUDA device).

    data_folder/data_file: .npz with 'data' of shape (T, N, F) -- the first feature is kept (utils.py:76);
    graph_csv: distance table with 'from', 'to' and the distance in the last column; id_file: optional
    text file of sensor ids (row order = node order); transform: None | 'standardize' | 'normalize'."""

    def __init__(self, data_folder, data_file, graph_csv, id_file=None, transform=None, device=None, verbose=True):
        import pandas as pd
        self.df = pd.read_csv(os.path.join(data_folder, graph_csv), index_col=None)
        if id_file is not None:
            sensor_id = np.loadtxt(os.path.join(data_folder, id_file), dtype=int)
            n_nodes = sensor_id.shape[0]
            sensor_dict = {int(sensor_id[k]): k for k in range(n_nodes)}
        else:
            n_nodes = max(max(self.df['from'].values), max(self.df['to'].values)) + 1
            sensor_dict = None
        n_edges, u_edges, u_distance = _u.physical_graph(self.df, sensor_dict)
        self.graph_info = {'n_nodes': int(n_nodes), 'n_edges': n_edges, 'u_edges': u_edges, 'u_dist': u_distance}
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        raw = np.load(os.path.join(data_folder, data_file), allow_pickle=False)['data'][..., :1]      # (T, N, 1)
        self.data = torch.from_numpy(np.ascontiguousarray(raw)).to(dev)
        if self.data.dtype not in _MG:
            self.data = self.data.double()
        self.transform = transform
        self.data_min, self.data_max, self.data_mean, self.data_std = series_stats(self.data)
        if verbose:
            sc = self.data_std.reshape(-1)
            print(f'[Metric=std] Disparity of each station: CV: {(sc.std() / sc.mean()).item():.4f}, '
                  f'Var: {sc.var().item():.4f}, PtP: {(sc.max() - sc.min()).item():.4f}')
        if transform == 'standardize':
            self._affine(self.data, self.data_mean, self.data_std, None, inverse=False)
        elif transform == 'normalize':
            self._affine(self.data, self.data_min, self.data_max, self.data_min, inverse=False)

    @staticmethod
    def _affine(x, shift, scale, scale_lo, inverse):
        n_steps, n_cols = _check_series(x)
        _lib.check(_lib.lib.mgadmm_series_affine(_ptr(x), n_steps, n_cols, _MG[x.dtype], _ptr(shift), _ptr(scale), _ptr(scale_lo),
                                                 int(inverse), _stream(x.device)))
        return x

    def recover_data(self, data):
        """Undo the transform on a (..., N, 1) tensor (utils.py:110-118); returns a new tensor on data's device."""
        if self.transform not in ('standardize', 'normalize'):
            return data
        x = data.to(device=self.data.device, dtype=self.data.dtype).contiguous().clone()
        flat = x.reshape(-1, *self.data.shape[1:])
        if self.transform == 'standardize':
            self._affine(flat, self.data_mean, self.data_std, None, inverse=True)
        else:
            self._affine(flat, self.data_min, self.data_max, self.data_min, inverse=True)
        return x.to(data.device)

    # --- the reference's single-window accessors (views of the resident series)
    def get_predict_data(self, index):
        return self.data[index:index + 24], self.data[index:index + 12]

    @staticmethod
    def interpolation_mask(shape, mask_rate=0.4, dtype=torch.float64):
        """The mask of get_interpolated_data (utils.py:128-129): `rand_like(x)` from torch's CPU generator seeded
        with 42, so it is the same for every window (the reference reseeds on each call)."""
        torch.manual_seed(42)
        return (torch.rand(shape, dtype=dtype) >= mask_rate).float()

    def get_interpolated_data(self, index, mask_rate=0.4):
        x = self.data[index:index + 24]
        mask = self.interpolation_mask(tuple(x.shape), mask_rate, x.dtype).to(x.device)
        return x, x * mask, mask

    # --- batches of windows (one kernel launch)
    def get_predict_batch(self, indices, T=24, t_in=12):
        """x (B, T, N, 1), y (B, t_in, N, 1) for a batch of window starts -- the solver's input layout."""
        x = gather_windows(self.data, indices, T)
        return x, x[:, :t_in].contiguous()

    def get_interpolated_batch(self, indices, mask_rate=0.4, T=24):
        """x, y = x * mask, mask, each (B, T, N, 1); every window carries the reference's seed-42 mask."""
        mask = self.interpolation_mask((T,) + tuple(self.data.shape[1:]), mask_rate, self.data.dtype)
        x = gather_windows(self.data, indices, T)
        y = gather_windows(self.data, indices, T, mask=mask)
        return x, y, mask.to(x.device).unsqueeze(0).expand(x.shape[0], *mask.shape).contiguous()
