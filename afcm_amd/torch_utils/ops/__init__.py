"""Host-side mirror of the reference's fused-op package (SG3OPS): same function names, keyword
defaults and error behaviour; the compute is the HIP library behind include/afcm_hip.h."""
