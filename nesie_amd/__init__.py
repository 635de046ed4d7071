"""nesie_amd -- MI355X (gfx950) native VoteNet/Nesie forward+backward hot path.

The package holds the hand-written HIP kernels (``csrc/`` -> ``libnesie_hip.so``,
C ABI in ``include/nesie_ops.h``), a host-side mirror of the reference's
``mmdet3d.ops`` operator surface (``nesie_amd.mmdet3d_ops``) and the detector
path built on it (``nesie_amd.votenet``).  There is no CPU path in here: ops
called on non-HIP tensors raise, and a missing ``libnesie_hip.so`` raises at
first use.
"""
__version__ = "0.1.0"
