"""Golden fixture for the generation loop from the REFERENCE's own `greedy_search`
(/root/reference/libra/models/libra/modeling_libra_utils.py:61) with its `ValidImageLogitsProcessor`
(/root/reference/libra/models/llama/modeling_llama_utils.py:23-76), driven exactly as the demo notebook drives it:
a LEFT-padded batch of two prompts,
   row 0:  pad pad pad pad BOS t t <img_gen>->BOI     -> generates an image (4 codes, EOI), the forced newline, then text
   row 1:  BOS BOI c c c c EOI t                      -> text continuation after an image (encoder signal on the 4 code rows)
The model is libra_tiny's (weights checked equal).  Stored: the prompt tensors, the generated sequences [Q,B,S+T] and the processed
per-step scores [T,Q,B,V+Vv].  Build-container only (imports /root/reference through ref_harness)."""
import importlib.util
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as rh  # noqa: E402
import make_golden_libra as mg  # noqa: E402


def _load_reference_generation():
    """The reference's generation utilities import beam-search modules the installed transformers no longer ships; they are
    dead for greedy_search / sample, so empty stand-in modules satisfy the import (nothing of them is executed)."""
    import transformers.generation as tg
    import transformers.generation.utils as tgu
    for name, syms in (("transformers.generation.beam_constraints", ("DisjunctiveConstraint", "PhrasalConstraint")),
                       ("transformers.generation.beam_search", ("BeamScorer", "BeamSearchScorer", "ConstrainedBeamSearchScorer"))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for s in syms:
                setattr(m, s, type(s, (), {}))
            sys.modules[name] = m
            setattr(tg, name.rsplit(".", 1)[1], m)
    # every other `from transformers.<module> import (names)` of the file: names the installed version dropped become empty
    # stand-in classes (greedy_search / sample never touch them)
    import ast
    import importlib
    with open(f"{rh.REF}/libra/models/libra/modeling_libra_utils.py") as f:
        tree = ast.parse(f.read())
    for node in tree.body:
        if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("transformers"):
            m = importlib.import_module(node.module)
            for alias in node.names:
                if not hasattr(m, alias.name):
                    stand_in = getattr(tgu, "GenerateDecoderOnlyOutput") if alias.name.endswith("Output") else type(alias.name, (), {})
                    setattr(m, alias.name, stand_in)
    spec = importlib.util.spec_from_file_location("_ref_modeling_libra_utils",
                                                  f"{rh.REF}/libra/models/libra/modeling_libra_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    spec2 = importlib.util.spec_from_file_location("_ref_modeling_llama_utils",
                                                   f"{rh.REF}/libra/models/llama/modeling_llama_utils.py")
    mod2 = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(mod2)
    return mod, mod2


def main():
    from helpers import load_golden, sub
    from make_golden import _save
    cfgm, ml, ll = rh.libra_modules()
    gen_utils, llama_utils = _load_reference_generation()
    cfg = cfgm.LibraConfig(**mg.TINY)
    torch.manual_seed(0)
    model = ml.LibraForCausalLM(cfg).eval()
    ref_w = sub(load_golden("libra_tiny.safetensors")[0], "w.")
    missing, unexpected = model.load_state_dict(ref_w, strict=False)
    assert not unexpected, unexpected
    V, L, Q = mg.TINY["vocab_size"], mg.TINY["max_vision_token_length"], 2
    boi, eoi = V + 16, V + 17
    g = torch.Generator().manual_seed(21)
    S, T = 8, 9
    ids = torch.zeros(Q, 2, S, dtype=torch.long)
    am = torch.ones(2, S, dtype=torch.long)
    vi = torch.full((2, S), L, dtype=torch.long)
    sig = torch.zeros(2, S, mg.TINY["contiguous_signal_size"])
    t0 = torch.randint(3, V - 2, (2,), generator=g)
    for q in range(Q):
        ids[q, 0] = torch.cat([torch.zeros(4, dtype=torch.long), torch.tensor([1]), t0, torch.tensor([boi])])
    am[0, :4] = 0
    vi[0, 7] = 0                                                       # <img_gen> -> BOI, vision index 0 (:275)
    t1 = torch.randint(3, V - 2, (1,), generator=g)
    for q in range(Q):
        ids[q, 1] = torch.cat([torch.tensor([1, boi]), V + torch.randint(0, 16, (4,), generator=g), torch.tensor([eoi]), t1])
    vi[1, 1:7] = torch.arange(6)
    sig[1, 2:6] = torch.randn(4, mg.TINY["contiguous_signal_size"], generator=g)
    proc = llama_utils.ValidImageLogitsProcessor(valid_image_token_length=4, boi_token_id=boi, eoi_token_id=eoi,
                                                 image_logits_offset=V, logits_size=V + mg.TINY["vision_vocab_size"])
    from transformers.generation.logits_process import LogitsProcessorList
    from transformers import GenerationConfig
    model.generation_config = GenerationConfig(pad_token_id=0, eos_token_id=2, output_scores=True, return_dict_in_generate=True)
    model._extract_past_from_model_output = lambda outputs, standardize_cache_format=False: outputs.past_key_values
    with torch.no_grad():
        out = gen_utils.BaseLibraPreTrainedModel.greedy_search(
            model, ids.clone(), logits_processor=LogitsProcessorList([proc]),
            stopping_criteria=lambda ids0, scores: ids0.shape[-1] >= S + T,       # (4.38's MaxLengthCriteria returned a bool)
            pad_token_id=0, eos_token_id=2,
            output_scores=True, return_dict_in_generate=True, attention_mask=am.clone(), vision_indices=vi.clone(),
            contiguous_signal=sig.clone(), use_cache=True)
    seq = out.sequences
    scores = torch.stack(out.scores)                                   # [T', Q, B, V + Vv]
    print("generated", seq.shape, "steps", scores.shape[0])
    print(seq[0])
    # structure check on row 0: 4 codes, EOI, newline
    new0 = seq[0, 0, S:]
    assert ((new0[:4] >= V) & (new0[:4] < V + 16)).all() and int(new0[4]) == eoi and int(new0[5]) == int(model.newline_token_id)
    t = {"in.input_ids": ids, "in.attention_mask": am, "in.vision_indices": vi, "in.signal": sig, "out.sequences": seq,
         "out.scores": scores}
    _save("libra_tiny_generate.safetensors", t, dict(cfg=mg.TINY, boi=boi, eoi=eoi, prompt_len=S, steps=int(scores.shape[0]),
                                                      valid_image_token_length=4, pad_token_id=0, eos_token_id=2,
                                                      newline_token_id=int(model.newline_token_id)))


if __name__ == "__main__":
    main()
