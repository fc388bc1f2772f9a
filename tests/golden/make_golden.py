"""Generates tests/golden/*.npz from the CPU oracle (fp64 evaluation of fp32 inputs).

The reference ships no golden vectors and cannot be imported here (TensorFlow is
absent), so these fixtures pin the ORACLE against regressions and give the GPU
tests committed expected outputs; they are not outputs of the TensorFlow code.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases as CS  # noqa: E402

SKIP = {"gated_n64", "residual_n64"}  # too large to commit; evaluated on the fly in the tests


def main():
    for name in CS.ATTN_CASES:
        if name in SKIP:
            continue
        inp, attrs, _ = CS.make_attn_case(name)
        out = CS.attn_oracle(inp, attrs)
        tree = {"in": {k: v for k, v in inp.items()},
                "out": {k: (None if v is None else v.float()) for k, v in out.items()}}
        CS.save_npz(os.path.join(CS.GOLDEN_DIR, f"attn_{name}.npz"), tree)
    for name in CS.BLOCK_CASES:
        if name in SKIP:
            continue
        inp, params, attrs, _ = CS.make_block_case(name)
        out = CS.block_oracle(inp, params, attrs)
        dparams = {k: (None if v is None else v.float()) for k, v in out.pop("dparams").items()}
        tree = {"in": inp, "params": params,
                "out": {k: v.float() for k, v in out.items()}, "dparams": dparams}
        CS.save_npz(os.path.join(CS.GOLDEN_DIR, f"block_{name}.npz"), tree)
    for name in CS.FFN_CASES:
        inp, params, c = CS.make_ffn_case(name)
        out = CS.ffn_oracle(inp, params, c)
        tree = {"in": inp, "params": params, "out": {"y": out["y"].float(), "dx": out["dx"].float()},
                "dparams": {k: v.float() for k, v in out["dparams"].items()}}
        CS.save_npz(os.path.join(CS.GOLDEN_DIR, f"ffn_{name}.npz"), tree)
    for name in CS.MODEL_CASES:
        inp, params, c = CS.make_model_case(name)
        out, dparams = CS.model_oracle(inp, params, c)
        keep = [k for k in dparams if dparams[k] is not None and
                (not k.startswith("layer") or k.startswith("layer0.") or "dense_qkv" in k)]   # a subset keeps the fixture small
        tree = {"in": inp, "out": {k: v.float() for k, v in out.items()},
                "dparams": {k: dparams[k].float() for k in keep}}
        CS.save_npz(os.path.join(CS.GOLDEN_DIR, f"model_{name}.npz"), tree)
    print("wrote", len(os.listdir(CS.GOLDEN_DIR)) - 1, "fixtures to", CS.GOLDEN_DIR)


if __name__ == "__main__":
    main()
