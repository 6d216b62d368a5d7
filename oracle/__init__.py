"""
CPU oracle for the ub-bonito CRF basecalling hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product package (xna_basecaller_amd) never does.  See xna_oracle.c for the
parity status ("parity unpinned" for the seqdist arithmetic; encoder pinned by
tests/golden/).
"""
from .oracle import (  # noqa: F401
    build,
    lib,
    crf_idx,
    decode,
    decode_logdomain,
    decode_scaled,
    pack,
    decode_batch,
    ctc_indices,
    ctc_logz,
    beam_search,
    log_beam_cut,
    conv1d_silu,
    lstm,
    linear_crf,
    encode,
    expf,
    logf,
    num_threads,
    set_num_threads,
    STATE_DICT_ORDER,
    state_dict_list,
)
