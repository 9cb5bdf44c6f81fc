"""MI355X implementation of the reference's src/infer_pack package (RVC synthesizer + NSF-HiFiGAN)."""
