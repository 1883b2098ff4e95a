"""oracle/amaxsum_oracle.c against the golden vectors generated from the reference's own amaxsum
computations (oracle/make_golden_amaxsum.py): runs everywhere, GPU box included."""
import os

import pytest

from amaxsum_common import check_golden, golden_files


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_reproduces_reference_amaxsum(path, oracle_built):
    from oracle.amaxsum_oracle import OracleAMaxSum
    check_golden(lambda g, p: OracleAMaxSum(g, p), path)
    assert len(golden_files()) >= 12
