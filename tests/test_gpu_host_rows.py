"""The host-side rows of SURVEY.md section 8 in the GPU-box record: the long-read haplotype filter (8f #4a), the realigner's host form, its
Python flow and de Bruijn consensus (8f #4b), and the host logic of the text seams carry no `gpu` mark - they need no device - so the driver's
`-m gpu` run on the MI355X box never executed them and the only GPU-box evidence for them were bench legs.  This wrapper runs those files, as
they are, in a child pytest on the box (its environment: the box's host cores, its libc / libstdc++, the .so files that travelled there) and
fails with the child's tail when any of them does."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

HOST_ROWS = ["test_haplotype_filtering.py", "test_realign.py", "test_realign_flow.py", "test_realign_batch.py", "test_host_logic.py",
             "test_cli_argv.py", "test_oracle_golden.py"]


def test_host_side_rows_pass_on_this_box():
    files = [os.path.join(ROOT, "tests", f) for f in HOST_ROWS]
    for f in files:
        assert os.path.exists(f), f
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail
