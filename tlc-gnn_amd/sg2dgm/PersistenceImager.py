"""Drop-in for the reference's sg2dgm/PersistenceImager.pyx (CSU-TDA PersistenceImages copy).

  linear_ramp :9-30, _norm_cdf :54-60, PersistenceImager.__init__ :207-242, _create_mesh :302-314, transform :352-403.

`transform` runs the HIP raster (`tlc_pi_raster`, include/tlcgnn.h).  The reference reaches only the isotropic
sigma=1 branch (:373-388) with the default ranges and the default linear ramp; anything else raises instead of
silently computing something different.
"""
import numpy as np

from .. import engine


def linear_ramp(birth, pers, low=0.0, high=1.0, start=0.0, end=1.0):
    """:9-30 (host helper kept for signature compatibility; the HIP raster applies the same ramp)."""
    n = birth.shape[0]
    w = np.zeros((n,))
    for i in range(n):
        if pers[i] < start:
            w[i] = low
        elif pers[i] > end:
            w[i] = high
        else:
            w[i] = (pers[i] - start) * (high - low) / (end - start) + low
    return w


def bvncdf(birth, pers, mu=None, sigma=None):  # marker for the default kernel (:32-51); evaluated on the GPU
    raise NotImplementedError("bvncdf is evaluated inside the HIP raster; the general (anisotropic) branch is not implemented")


class PersistenceImager:
    def __init__(self, birth_range=None, pers_range=None, pixel_size=None, resolution=5,
                 weight=linear_ramp, weight_params=None, kernel=bvncdf, kernel_params=None):
        if birth_range is None:
            birth_range = (0.0, 1.0)
        if pers_range is None:
            pers_range = (0.0, 1.0)
        self._resolution = (resolution, resolution)
        if pixel_size is None:
            pixel_size = np.min([pers_range[1] - pers_range[0], birth_range[1] - birth_range[0]]) / resolution
        if weight_params is None:
            weight_params = {}
        if kernel_params is None:
            kernel_params = {'sigma': np.array([[1.0, 0.0], [0.0, 1.0]])}
        sigma = np.asarray(kernel_params.get('sigma'), dtype=np.float64)
        if (tuple(birth_range) != (0.0, 1.0) or tuple(pers_range) != (0.0, 1.0) or weight is not linear_ramp
                or weight_params or kernel is not bvncdf or abs(pixel_size - 1.0 / resolution) > 1e-15
                or not np.array_equal(sigma, np.eye(2)) or not (1 <= resolution <= 8)):
            raise NotImplementedError("PersistenceImager: only the configuration the TLC-GNN pipeline uses is implemented "
                                      "(ranges [0,1], sigma=I, linear_ramp defaults, resolution 1..8)")
        self.weight, self.weight_params, self.kernel, self.kernel_params = weight, weight_params, kernel, kernel_params
        self._pixel_size = pixel_size
        self._birth_range, self._pers_range = tuple(birth_range), tuple(pers_range)
        self._width = birth_range[1] - birth_range[0]
        self._height = pers_range[1] - pers_range[0]
        self._create_mesh()

    resolution = property(lambda self: self._resolution)
    pixel_size = property(lambda self: self._pixel_size)
    birth_range = property(lambda self: self._birth_range)
    pers_range = property(lambda self: self._pers_range)
    width = property(lambda self: self._width)
    height = property(lambda self: self._height)

    def _create_mesh(self):
        # :302-314 (db = dp = 0 for the supported configuration)
        self._bpnts = np.array(np.linspace(self._birth_range[0], self._birth_range[1] + self._pixel_size,
                                           self._resolution[0] + 1, endpoint=False, dtype=np.float64))
        self._ppnts = np.array(np.linspace(self._pers_range[0], self._pers_range[1] + self._pixel_size,
                                           self._resolution[1] + 1, endpoint=False, dtype=np.float64))

    def transform(self, pers_dgm, skew=True):
        """:352-403.  pers_dgm: (N,2) birth-death pairs -> ndarray (res,res), [birth_bin, pers_bin]."""
        if not skew:
            raise NotImplementedError("transform(skew=False) is not implemented on the HIP path")
        import torch
        d = np.ascontiguousarray(np.asarray(pers_dgm, dtype=np.float64).reshape(-1, 2))
        res = self._resolution[0]
        offs = torch.tensor([0, d.shape[0]], dtype=torch.int64, device="cuda")
        out = engine.pi_raster(offs, torch.from_numpy(d).cuda(), res)
        return out[0].cpu().numpy().reshape(self._resolution)

    def transform_batch(self, diagrams):
        """Many diagrams in one launch (one wavefront per diagram); returns ndarray [B, res, res]."""
        import torch
        res = self._resolution[0]
        lens = [len(x) for x in diagrams]
        offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64, device="cuda")
        pts = np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1, 2) for x in diagrams]) if sum(lens) else np.zeros((0, 2))
        out = engine.pi_raster(offs, torch.from_numpy(np.ascontiguousarray(pts)).cuda(), res)
        return out.cpu().numpy().reshape(len(diagrams), res, res)
