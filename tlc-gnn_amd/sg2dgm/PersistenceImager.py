"""Drop-in for the reference's sg2dgm/PersistenceImager.pyx (CSU-TDA PersistenceImages copy).

  linear_ramp :9-30, _norm_cdf :54-60, PersistenceImager.__init__ :207-242, _create_mesh :302-314, transform :352-403.

`transform` runs the HIP raster (`tlc_pi_raster`, include/tlcgnn.h).  The reference reaches only the isotropic
sigma=1 branch (:373-388) with the default ranges and the default linear ramp; anything else raises instead of
silently computing something different.
"""
import numpy as np

from .. import engine


def linear_ramp(birth, pers, low=0.0, high=1.0, start=0.0, end=1.0):
    """Weight of a (birth, persistence) pair (:9-30): `low` below `start`, `high` above `end`, the straight line between.
    Host-side twin of the ramp the HIP raster applies; vectorised (`birth` only fixes the length)."""
    pers = np.asarray(pers, dtype=np.float64)
    line = (pers - start) * (high - low) / (end - start) + low
    return np.where(pers < start, low, np.where(pers > end, high, line))


def bvncdf(birth, pers, mu=None, sigma=None):  # marker for the default kernel (:32-51); evaluated on the GPU
    raise NotImplementedError("bvncdf is evaluated inside the HIP raster; the general (anisotropic) branch is not implemented")


_UNIT = (0.0, 1.0)


class PersistenceImager:
    """Constructor signature of :207-208.  Only the configuration the pipeline instantiates is accepted -- unit birth and
    persistence ranges, square pixels of 1/resolution, sigma = I, the default ramp -- so the state below is derived from
    `resolution` alone; anything else raises."""

    def __init__(self, birth_range=None, pers_range=None, pixel_size=None, resolution=5,
                 weight=linear_ramp, weight_params=None, kernel=bvncdf, kernel_params=None):
        sigma = np.eye(2) if kernel_params is None else np.asarray(kernel_params.get('sigma'), dtype=np.float64)
        supported = (
            tuple(birth_range or _UNIT) == _UNIT and tuple(pers_range or _UNIT) == _UNIT
            and weight is linear_ramp and not weight_params and kernel is bvncdf
            and isinstance(resolution, (int, np.integer)) and 1 <= resolution <= 8
            and (pixel_size is None or abs(pixel_size - 1.0 / resolution) <= 1e-15)
            and sigma.shape == (2, 2) and np.array_equal(sigma, np.eye(2)))
        if not supported:
            raise NotImplementedError("PersistenceImager: only the configuration the TLC-GNN pipeline uses is implemented "
                                      "(ranges [0,1], sigma=I, linear_ramp defaults, resolution 1..8)")
        res = int(resolution)
        self.weight, self.weight_params = weight, {}
        self.kernel, self.kernel_params = kernel, {'sigma': sigma}
        self._resolution = (res, res)
        self._pixel_size = 1.0 / res
        self._birth_range = self._pers_range = _UNIT
        self._width = self._height = 1.0
        # pixel boundaries (:302-314 with zero padding): res + 1 points, spacing (1 + pixel) / (res + 1)
        edges = (1.0 + self._pixel_size) / (res + 1) * np.arange(res + 1, dtype=np.float64)
        self._bpnts, self._ppnts = edges, edges.copy()

    resolution = property(lambda self: self._resolution)
    pixel_size = property(lambda self: self._pixel_size)
    birth_range = property(lambda self: self._birth_range)
    pers_range = property(lambda self: self._pers_range)
    width = property(lambda self: self._width)
    height = property(lambda self: self._height)

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
