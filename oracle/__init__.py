"""CPU oracle for the stereo-pair -> disparity-map path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product package (stereo_matching_cuda_amd) never does.  See
oracle/smx_oracle.c for the restatement and how it is pinned.
"""
from .oracle import *  # noqa: F401,F403
