"""flowcompare_amd — MI355X-native forward log-prob engine for FlowCompare's conditional normalizing flow.

Host-side mirror of the reference module API (initialize_flow / inner_loop / Flow.log_prob) over a C-ABI HIP
library (libfcflow.so, include/fcflow.h).  There is no PyTorch/CPU fallback: every compute entry point raises if
the HIP library is missing.
"""
from .config import config_loader, named_config  # noqa: F401
from .model_initialization import initialize_flow, inner_loop, make_sample, save_flow, load_flow  # noqa: F401

__all__ = ["config_loader", "named_config", "initialize_flow", "inner_loop", "make_sample", "save_flow", "load_flow"]
