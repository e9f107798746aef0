"""stmask_amd -- MI355X-native (gfx950) implementation of STMask's per-frame inference hot path.

Layout:
  csrc/                    hand-written HIP kernels + the C ABI (include/stmask_hip.h) -> libstmask_hip.so
  _lib.py, ops.py          ctypes binding and tensor-level wrappers (no CPU fallback)
  dcn_v2.py, mmcv_ops.py, spatial_correlation_sampler.py
                           drop-ins for the reference's third-party imports (shims/ exposes them by name)
  layers/, backbone.py, model.py, config.py
                           host-side mirror of the reference's layer API (same names, state-dict keys)
  pipeline.py, dist.py     batched clip pipeline and clip sharding over GPUs (RCCL all-gather of detections)
"""
__version__ = "0.1.0"
