import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The CLIP BPE merge table is third-party data the repo does not carry.  In the build container the tokenizer tests borrow the copy that
# sits next to the reference checkout (test infrastructure only: the product looks at $SEMABS_BPE_VOCAB / its assets dir / ~/.cache/clip);
# on the GPU box the path does not exist and those tests use the committed token-id fixtures instead.
_BPE = "/root/reference/CLIP/clip/bpe_simple_vocab_16e6.txt.gz"
if "SEMABS_BPE_VOCAB" not in os.environ and os.path.isfile(_BPE):
    os.environ["SEMABS_BPE_VOCAB"] = _BPE


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


def sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
