"""configs[4] at FULL size inside the GPU suite (run with `-m gpu`).  A module of its own: the bench runs as a child process and needs the
whole HBM, which the module-scoped contexts of the other large tests hold while their modules run (their arenas go back when they close)."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(1500)
def test_configs4_at_full_size_against_the_reference_digests():
    """BASELINE configs[4] at its FULL 20 M reads inside the GPU suite, not only in the builder's own bench runs —
    `python bench.py --config c5` (5 protein-guided + 5 nucleotide iterations on 20 M reads / 35.3 M ORFs, one untimed traversal digested in HBM)
    must reproduce tests/golden/c5_chain_digests.json, digests the unmodified `penguin` binaries produce for the same reads
    (profiles/r05_headline_pin_reference.txt: 15 of 15 DBs)."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    gold = json.load(open(os.path.join(HERE, "golden", "c5_chain_digests.json")))
    assert gold["pairs"] == 10000000 and len(gold["digests"]) == 10
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "c5", "--steps", "10", "--warmup", "0", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1400)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["verify"]["match"] is True and line["verify"]["step_digests"] == gold["digests"], line["verify"]
    assert line["roofline"]["furthest_below"] is not None and line["config"]["candidate_overlaps"] > 5e8

