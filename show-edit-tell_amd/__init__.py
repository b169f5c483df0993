"""show-edit-tell decode path, MI355X-native (gfx950 HIP kernels behind the reference's nn.Module API).

Sub-modules (imported lazily so that `synth` is usable without torch / the HIP library):
  synth            deterministic synthetic inputs + weights
  _lib             ctypes binding of the C-ABI library csrc/libset_hip.so (fails loudly if absent)
  build            hipcc build of the library for gfx950
  editnet          DecoderC with the XE forward           (reference editnet.py)
  editnet_rl       DecoderC with the greedy/sampling forward (reference editnet_rl.py)
  editnet_adaptive DecoderC for 10-100 adaptive regions   (reference adaptive_features/editnet_adaptive.py)
  dcnet, dcnet_rl  DAE / DAEWithAR                        (reference dcnet.py, dcnet_rl.py)
"""
__version__ = "0.1.0"
