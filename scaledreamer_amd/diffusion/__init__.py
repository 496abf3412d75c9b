"""SD-v2.1 / MVDream UNet and VAE-encoder execution for the ASD guidance step."""
