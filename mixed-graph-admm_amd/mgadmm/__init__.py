"""mgadmm -- MI355X-native mixed-graph ADMM solver (drop-in for JiQi-da/Mixed-Graph-ADMM's hot path).

    from mgadmm.ADMM import ADMM_algorithm, initial_guess, initial_interpolation
    from mgadmm.utils import k_nearest_neighbors, connect_list, ...
    from mgadmm.CG_script import conjugate_gradient

Importing the package loads libmgadmm.so (HIP kernels + C ABI) and fails loudly when it has not
been built; there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises OSError when the HIP extension is missing)
from . import utils  # noqa: F401
from .ADMM import ADMM_algorithm, initial_guess, initial_interpolation  # noqa: F401
from .CG_script import conjugate_gradient  # noqa: F401
from .dist import shard_bounds, sharded_solve  # noqa: F401
from . import gpu_graph  # noqa: F401
from .dataset import TrafficDataset  # noqa: F401

__version__ = _lib.version()
