"""oracle -- CPU restatement of the RASR front-end / scorers.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (rasr_amd) never does.  See oracle/orc.h for what is restated and how
each piece is pinned against the reference.
"""
from .binding import (Oracle, load_ref, MfccCfg, build_oracle, OracleMfcc, OracleGmm, oracle_ffnn_score)  # noqa: F401
