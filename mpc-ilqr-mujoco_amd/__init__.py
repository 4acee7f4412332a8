"""MI355X-native batched iLQR for the Unitree H1 -- host-side Python mirror of the reference's
iLQR / MPC interface (include/ilqr/ilqr.hpp, include/ilqr/mpc.hpp) over the C-ABI HIP library."""
from . import scenario  # noqa: F401
