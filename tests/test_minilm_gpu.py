"""GPU: caption embeddings through the C ABI (cap_embed_text) against the HF-captured goldens and the restatement."""
import numpy as np
import pytest
import torch

from test_minilm_cpu import load

pytestmark = pytest.mark.gpu


def _engine(arch, dtype, batch, L):
    from embodied_captioning_amd.engine import TextEncoderEngine
    return TextEncoderEngine(arch, dtype=dtype, max_batch=batch, max_len=L)


@pytest.mark.parametrize("name", ["minilm_tiny", "minilm_base"])
@pytest.mark.parametrize("dtype,tol,cos", [("f32", 5e-6, 1 - 1e-6), ("bf16", 2e-2, 0.999)])
def test_embeddings_match_hf_golden(name, dtype, tol, cos):
    from embodied_captioning_amd.weights import procedural_minilm_state_dict
    g, meta, arch = load(name)
    eng = _engine(arch, dtype, meta["batch"], meta["L"])
    eng.load_state_dict(procedural_minilm_state_dict(arch, meta["seed"]))
    out = eng.embed(torch.from_numpy(g["ids"]), torch.from_numpy(g["lens"])).cpu().numpy()
    ref = g["embeddings"]
    assert np.abs(out - ref).max() < tol, np.abs(out - ref).max()
    assert ((out * ref).sum(1) > cos).all()
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
    eng.close()


def test_ragged_batches_and_padding_invariance_fp32():
    """Caption-sized workload (256 sentences, 3..24 tokens): identical rows whether embedded alone, in the batch, or with
    other padding ids; checked against the restatement."""
    from embodied_captioning_amd.config import MiniLMArch
    from embodied_captioning_amd.weights import procedural_minilm_state_dict, synthetic_token_batch
    from oracle import minilm_ref as R
    arch = MiniLMArch()
    sd = procedural_minilm_state_dict(arch, 2)
    ids, lens = synthetic_token_batch(arch, 256, 24, 2)
    eng = _engine(arch, "f32", 256, 24)
    eng.load_state_dict(sd)
    full = eng.embed(ids, lens).cpu()
    ref = R.encode_tokens(sd, arch, ids[:32], lens[:32])
    assert (full[:32] - ref).abs().max().item() < 5e-6
    junk = ids.clone()
    for b in range(256):
        junk[b, int(lens[b]):] = 11
    assert torch.equal(eng.embed(junk, lens).cpu(), full)
    n = int(lens[5])
    alone = eng.embed(ids[5:6, :n], lens[5:6]).cpu()
    assert (alone[0] - full[5]).abs().max().item() < 2e-6
    eng.close()


def test_sentence_encoder_surface_and_errors():
    from embodied_captioning_amd._native import CaptionerHipError
    from embodied_captioning_amd.captioner.sentence_encoder import SentenceEncoder
    enc = SentenceEncoder("procedural-minilm-tiny:3", dtype="f32", batch_size=4).to("cuda:0")
    assert enc.get_sentence_embedding_dimension() == enc.arch.hidden
    rows = [[1, 9, 8, 2], [1, 5, 2], [1, 17, 33, 21, 4, 2], [1, 2], [1, 40, 41, 2], [1, 9, 8, 2]]
    emb = enc.encode_ids(rows)                                         # 6 rows through batches of 4, sorted by length inside
    assert emb.shape == (6, enc.arch.hidden) and torch.equal(emb[0], emb[5])
    with pytest.raises(RuntimeError):
        enc.encode("a caption")                                        # procedural weights have no vocabulary
    with pytest.raises(ValueError):
        enc.engine.embed(torch.tensor([[1, 400, 2]]), torch.tensor([3]))   # id outside the vocabulary
    with pytest.raises(CaptionerHipError):
        enc.engine.embed(torch.ones(5, 3, dtype=torch.int32), torch.full((5,), 3, dtype=torch.int32))   # batch > capacity
    with pytest.raises(RuntimeError):
        SentenceEncoder("all-MiniLM-L6-v2-not-here")


def test_embeddings_from_caption_tokens_match_row_form():
    from embodied_captioning_amd.captioner.sentence_encoder import SentenceEncoder
    enc = SentenceEncoder("procedural-minilm-tiny:3", dtype="f32", batch_size=8)
    a = enc.arch
    eos, bos = 102, 290
    seq = torch.tensor([[bos, 7, 9, 11, eos, 0, 0], [bos, 5, eos, 0, 0, 0, 0], [bos, 8, 9, 10, 12, 13, 14]], dtype=torch.int32)
    lens = torch.tensor([5, 3, 7], dtype=torch.int32)
    got = enc.encode_caption_tokens(seq, lens, eos)
    want = enc.encode_ids([[a.cls, 7, 9, 11, a.sep], [a.cls, 5, a.sep], [a.cls, 8, 9, 10, 12, 13, 14, a.sep]])
    assert torch.equal(got, want)
