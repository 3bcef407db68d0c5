"""Host-side mirror of the reference `lvdm` interface for the denoising hot path.

Each module here carries the class names, constructor kwargs, call signatures and
parameter names of its reference counterpart (file:line cited per module), but its
forward runs on the hand-written HIP operators in `tooncrafter_amd.ops`.
`tooncrafter_amd.dropin.install()` publishes these modules under the reference's
dotted paths (`lvdm.models.ddpm3d`, ...) so `configs/inference_512_v1.0.yaml`
resolves to them unmodified.
"""
