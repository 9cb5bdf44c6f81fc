"""aicovergen_amd -- MI355X (gfx950) native hot path of AICoverGen: MDX-Net separation and RVC voice
conversion behind the reference's own Python call surface (mdx.run_mdx, vc_infer_pipeline.VC, ...).
All arithmetic runs in hand-written HIP kernels (csrc/) bound through the C ABI in include/aicg.h."""
__version__ = "0.1.0"
