"""`import dllogger` for the reference's entry points: the DLLL-line writer of deeplearningexamples_amd.utils.dllogger."""
from deeplearningexamples_amd.utils.dllogger import *            # noqa: F401,F403
from deeplearningexamples_amd.utils.dllogger import Verbosity, StdOutBackend, JSONStreamBackend, init, log, metadata, flush  # noqa: F401
