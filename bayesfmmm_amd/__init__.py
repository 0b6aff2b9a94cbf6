"""bayesfmmm_amd: MI355X-native Gibbs sampler for the functional / multivariate mixed-membership
models of ndmarco/BayesFMMM (hot path only).  The compute lives in
libbfmmm_hip.so behind the C ABI of include/bfmmm.h; this package is the host-side mirror of the
reference's entry points over that ABI."""
from . import _lib  # noqa: F401
from .sampler import (MODEL_FUNCTIONAL, MODEL_MULTIVARIATE, SWEEP_NU_Z, SWEEP_THETA, SWEEP_WARM,  # noqa: F401
                      Sampler, default_config)
