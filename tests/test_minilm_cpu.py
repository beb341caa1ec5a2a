"""CPU: the sentence-encoder restatement (oracle/minilm_ref.py) against vectors captured from the real HF BertModel."""
import json
import os

import numpy as np
import pytest
import torch

from embodied_captioning_amd.config import MiniLMArch
from embodied_captioning_amd.weights import procedural_minilm_state_dict, synthetic_token_batch
from oracle import minilm_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    arch = MiniLMArch(**meta["arch"])
    return g, meta, arch


@pytest.mark.parametrize("name", ["minilm_tiny", "minilm_base"])
def test_restatement_matches_hf_golden(name):
    g, meta, arch = load(name)
    sd = procedural_minilm_state_dict(arch, meta["seed"])
    ids, lens = synthetic_token_batch(arch, meta["batch"], meta["L"], meta["seed"])
    assert np.array_equal(ids.numpy(), g["ids"]) and np.array_equal(lens.numpy(), g["lens"])     # inputs are reproducible
    emb = R.encode_tokens(sd, arch, ids, lens)
    assert np.abs(emb.numpy() - g["embeddings"]).max() < 2e-6
    assert np.allclose(np.linalg.norm(emb.numpy(), axis=1), 1.0, atol=1e-6)


def test_padding_and_batch_invariance():
    """A sentence's embedding does not depend on the padding after it or on its batch neighbours."""
    arch = MiniLMArch.tiny()
    sd = procedural_minilm_state_dict(arch, 1)
    ids, lens = synthetic_token_batch(arch, 5, 10, 1)
    full = R.encode_tokens(sd, arch, ids, lens)
    for b in range(5):
        n = int(lens[b])
        alone = R.encode_tokens(sd, arch, ids[b:b + 1, :n], lens[b:b + 1])
        assert (alone[0] - full[b]).abs().max().item() < 1e-6
    junk = ids.clone()
    for b in range(5):
        junk[b, int(lens[b]):] = 7                                      # different padding ids
    assert (R.encode_tokens(sd, arch, junk, lens) - full).abs().max().item() < 1e-6
