"""salsa_amd -- MI355X-native SALSA / SALSA-Lite spatial-audio feature extraction (hand-written HIP for gfx950
behind the C ABI of include/salsa_hip.h).  See DESIGN.md and INTEGRATION.md."""
__version__ = '0.1.0'
