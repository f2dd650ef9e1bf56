"""discontinuum_amd -- MI355X-native exact-GP marginal-likelihood engine behind discontinuum's
``fit / predict`` surface.  The arithmetic lives in ``libdgp_hip.so`` (hand-written HIP for gfx950);
see DESIGN.md and include/dgp_hip.h."""
__version__ = "0.1.0"
