"""Generation path on device: the product's greedy_search / sample over [Q,B,S] ids with left-padded prompts
against the fixture produced by the reference's own greedy_search (tests/golden/make_golden_libra_generate.py)."""
import pytest
import torch

from helpers import load_golden, parity_report, rel_err, sub

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _model():
    from libra_amd.libra import LibraConfig, LibraForCausalLM
    t, meta = load_golden("libra_tiny_generate.safetensors")
    w = sub(load_golden("libra_tiny.safetensors")[0], "w.")
    m = LibraForCausalLM(LibraConfig(**meta["cfg"]))
    m.load_state_dict(w, strict=True)
    return m.to(BF).cuda().eval(), t, meta


def _proc(meta):
    from libra_amd.libra.generation import ValidImageLogitsProcessor
    c = meta["cfg"]
    return ValidImageLogitsProcessor(meta["valid_image_token_length"], meta["boi"], meta["eoi"], c["vocab_size"],
                                     c["vocab_size"] + c["vision_vocab_size"])


def test_greedy_search_vs_reference_fixture():
    m, t, meta = _model()
    S, T = meta["prompt_len"], meta["steps"]
    out = m.greedy_search(t["in.input_ids"].cuda(), logits_processor=[_proc(meta)], max_length=S + T,
                          pad_token_id=meta["pad_token_id"], eos_token_id=meta["eos_token_id"], output_scores=True,
                          return_dict_in_generate=True, attention_mask=t["in.attention_mask"].cuda(),
                          vision_indices=t["in.vision_indices"].cuda(), contiguous_signal=t["in.signal"].to(BF).cuda())
    seq, ref = out.sequences.cpu(), t["out.sequences"]
    assert seq.shape == ref.shape
    scores, rscores = torch.stack(out.scores).float().cpu(), t["out.scores"]
    worst = 0.0
    for step in range(T):
        # while both runs are on the same path the processed scores must agree (same -inf / +inf pattern, close values)
        a, b = scores[step], rscores[step]
        assert torch.equal(torch.isfinite(a), torch.isfinite(b)) and torch.equal(torch.isposinf(a), torch.isposinf(b)), step
        fin = torch.isfinite(b)
        worst = max(worst, rel_err(a[fin], b[fin]))
        assert rel_err(a[fin], b[fin]) < 3e-2, (step, rel_err(a[fin], b[fin]))
        same = seq[:, :, S + step] == ref[:, :, S + step]
        if not bool(same.all()):
            # bf16 against the reference's fp32 run: a different argmax is admissible only at a near-tie of the reference
            top2 = torch.topk(torch.nan_to_num(b, neginf=-1e9, posinf=1e9), 2, dim=-1).values
            margin = (top2[..., 0] - top2[..., 1])[~same]
            assert float(margin.max()) < 0.05 * float(b[fin].abs().max()), (step, margin)
            parity_report(f"[f1 greedy_search] diverged from the reference at step {step} on a near-tie (margin {float(margin.max()):.3e})")
            break
    else:
        assert torch.equal(seq, ref)
    new0 = seq[0, 0, S:]
    V = meta["cfg"]["vocab_size"]
    assert bool(((new0[:4] >= V) & (new0[:4] < V + 16)).all()) and int(new0[4]) == meta["eoi"] and int(new0[5]) == meta["newline_token_id"]
    parity_report(f"[f1 greedy_search, left-padded batch, tiny] {T} steps: sequences {'equal' if torch.equal(seq, ref) else 'equal up to a near-tie'}"
                  f" to the reference's greedy_search; worst per-step score rel err {worst:.3e}")


def test_left_padded_prompt_equals_unpadded_prompt():
    """Row 0 of the fixture batch is left-padded by 4: its prefill logits and decode steps must equal the same prompt run alone
    without padding (pad keys masked by kv_start, positions = cumsum - 1), and graphs / eager decode agree."""
    m, t, meta = _model()
    ids, am, vi = t["in.input_ids"].cuda(), t["in.attention_mask"].cuda(), t["in.vision_indices"].cuda()
    sig = t["in.signal"].to(BF).cuda()
    a = m(input_ids=ids, attention_mask=am, vision_indices=vi, contiguous_signal=sig, use_cache=True)
    b = m(input_ids=ids[:, :1, 4:], attention_mask=am[:1, 4:], vision_indices=vi[:1, 4:], contiguous_signal=sig[:1, 4:], use_cache=True)
    la, lb = a.logits[:, 0, 4:].float(), b.logits[:, 0].float()
    assert torch.equal(torch.isfinite(la), torch.isfinite(lb))
    fin = torch.isfinite(lb)
    assert rel_err(la[fin].cpu(), lb[fin].cpu()) < 1e-2
    assert a.past_key_values.start is not None and a.past_key_values.start.tolist() == [4, 0] and b.past_key_values.start is None
    nxt = torch.full((2, 2, 1), meta["cfg"]["vocab_size"] + 3, dtype=torch.long, device="cuda")     # a code token for both rows
    nxt[:, 1] = 5                                                                                   # row 1 continues with text
    vin = torch.tensor([[1], [meta["cfg"]["max_vision_token_length"]]], device="cuda")
    a2 = m(input_ids=nxt, vision_indices=vin, past_key_values=a.past_key_values, use_cache=True)
    b2 = m(input_ids=nxt[:, :1], vision_indices=vin[:1], past_key_values=b.past_key_values, use_cache=True)
    fin = torch.isfinite(b2.logits[:, 0].float())
    assert rel_err(a2.logits[:, 0].float()[fin].cpu(), b2.logits[:, 0].float()[fin].cpu()) < 1e-2


def test_sample_is_well_formed_and_topk1_equals_greedy():
    m, t, meta = _model()
    S, T = meta["prompt_len"], meta["steps"]
    kw = dict(attention_mask=t["in.attention_mask"].cuda(), vision_indices=t["in.vision_indices"].cuda(),
              contiguous_signal=t["in.signal"].to(BF).cuda(), pad_token_id=0, eos_token_id=2)
    greedy = m.greedy_search(t["in.input_ids"].cuda(), logits_processor=[_proc(meta)], max_length=S + T, **kw)
    top1 = m.generate(t["in.input_ids"].cuda(), do_sample=True, top_k=1, logits_processor=[_proc(meta)], max_length=S + T, **kw)
    assert torch.equal(greedy, top1)
    gen = torch.Generator(device="cuda").manual_seed(0)
    smp = m.sample(t["in.input_ids"].cuda(), logits_processor=[_proc(meta)], max_length=S + T, generator=gen, **kw)
    V = meta["cfg"]["vocab_size"]
    new0 = smp[:, 0, S:]
    assert bool(((new0[:, :4] >= V) & (new0[:, :4] < V + 16)).all())                   # 4 codes in BOTH codebooks
    assert bool((new0[:, 4] == meta["eoi"]).all()) and bool((new0[:, 5] == meta["newline_token_id"]).all())
    assert smp.shape == greedy.shape


def test_kv_cache_capacity_and_graph_bound():
    from libra_amd import decoder_engine as DE
    m, t, meta = _model()
    kw = dict(attention_mask=t["in.attention_mask"].cuda(), vision_indices=t["in.vision_indices"].cuda(),
              contiguous_signal=t["in.signal"].to(BF).cuda())
    out = m(input_ids=t["in.input_ids"].cuda(), use_cache=True, max_cache_len=12, **kw)
    cache = out.past_key_values
    assert cache.capacity == 12 and cache.layers[0][0].shape[1] == 12                  # sized from max_length, not max_position_embeddings
    V, L = meta["cfg"]["vocab_size"], meta["cfg"]["max_vision_token_length"]
    for step in range(4):
        ids = torch.full((2, 2, 1), 7 + step, dtype=torch.long, device="cuda")
        m(input_ids=ids, vision_indices=torch.full((2, 1), L, device="cuda"), past_key_values=cache, use_cache=True)
    assert cache.length == 12 and len(cache.graphs) <= DE.MAX_DECODE_GRAPHS
    with pytest.raises(ValueError, match="full"):
        m(input_ids=ids, vision_indices=torch.full((2, 1), L, device="cuda"), past_key_values=cache, use_cache=True)
