"""gast_hip: MI355X (gfx950) kernels + ctypes binding + host plan for the GAST-Net spatio-temporal hot path."""
