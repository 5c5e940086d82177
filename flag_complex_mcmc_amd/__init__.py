"""flag_complex_mcmc_amd -- MI355X-native edge-flip MCMC for directed flag complexes.

Drop-in for the hot path of TheJonny/flag-complex-mcmc (the `sample` binary's
step loop and the simplex counter it calls).  See DESIGN.md and include/fcm.h.
"""
from ._ffi import FcmError, device_count, lib, LIB_PATH  # noqa: F401
from .api import (Graph, Bounds, MCMCSampler, MultiDeviceSampler, State, Transition, BitOutput, initialize_new_sampler,  # noqa: F401
                  read_flag_file, save_flag_file, count_unweighted, default_sample_distance,
                  MOVE_DISTRIBUTION, MOVE_DISTRIBUTION_SIMPLE)
from . import graphs  # noqa: F401
