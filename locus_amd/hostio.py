"""Moving the p x d matrices between the host (R's layout: float64, column-major) and a rank's
[p][d_loc] device planes without extra host passes."""
import numpy as np


def columns_to_device(a, cols, device):
    """Columns `cols` (slice or index array over the responses) of a p x d matrix as a row-major
    [p][len(cols)] float64 tensor on `device`.  `a` is a host array, or a
    locus_amd.prepare.DrawnMatrix, which draws those columns on the device itself."""
    import torch
    if hasattr(a, "columns_on"):
        return a.columns_on(cols, device)
    a = np.asarray(a, dtype=np.float64)
    return torch.from_numpy(np.ascontiguousarray(a[:, cols])).to(device)


def _to_host(t):
    """Device tensor -> host tensor through a page-locked destination: 57 GB/s measured on the
    MI355X box against 4 GB/s into pageable memory (scripts/perf_d2h.py).  The numpy array handed
    to the caller lives in that page-locked block."""
    import torch
    if t.device.type == "cpu":                 # test backend: never alias the engine's buffers
        return t.clone()
    try:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:                       # no page-locked memory left: plain copy
        return t.cpu()
    h.copy_(t)
    return h


def to_host_F(t):
    """Row-major [rows][cols] device tensor -> Fortran-ordered numpy (rows x cols); the transpose
    runs on the device, the host sees one copy."""
    if t.dim() != 2:
        return _to_host(t.contiguous()).numpy().copy()
    return _to_host(t.t().contiguous()).numpy().T


class LazyFields(dict):
    """Fields of a driver's result that are downloaded on first use: locus() only needs gam_vb and
    beta_vb unless full_output is set (R/locus_core.R:242-252)."""

    def __init__(self, **makers):
        super().__init__()
        self._makers = makers

    def __missing__(self, key):
        v = self[key] = self._makers[key]()
        return v

    def all(self):
        for k in self._makers:
            self[k]
        return dict(self)
