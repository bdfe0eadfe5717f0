"""oakink2_tamf_amd - MI355X-native MF-MDM denoiser + DDPM sampler behind OakInk2-TaMF's call contracts.

Layout mirrors the reference package for the hot path only:
  model/interaction_segment_mdm.py  InterationSegmentMDM   (reference: model/interaction_segment_mdm.py)
  model/segment_refine_model.py     SegmentRefineModel     (reference: model/segment_refine_model.py)
  model/diffusion_util.py           create_gaussian_diffusion (reference: model/diffusion_util.py)
  model/diffusion/gaussian_diffusion.py  schedule tables + p_sample_loop (reference: same path)
  launch/sample.py, sample_refine.py, upkeep.py   sample.sh / sample_refine.sh-compatible CLIs (reference: launch/, dev_fn/upkeep)
  geometry.py                       pose decode, hand->object distance, vertex normals, contact / SIV kernels
  hip_backend.py                    ctypes binding of include/tamf_hip.h (libtamf_hip.so, gfx950)
  shard.py                          clip sharding across ranks + RCCL result gather
The compute path is the HIP library; there is no CPU fallback (importing works without a GPU,
running does not).
"""
__version__ = "0.1.0"
