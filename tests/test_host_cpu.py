"""Host-side mirror of the reference interface (no GPU needed): Keras variable names, weight
loading, constructor errors."""
import numpy as np
import pytest
import torch


def test_layer_stack_keras_names_and_weight_loading():
    from egt_amd import EGTLayerStack
    st = EGTLayerStack(model_height=2, model_width=64, edge_width=64, num_heads=8)
    named = st.keras_named_parameters()
    # the variable names the reference's Keras layers create (graph_xformer_model_base.py:109-218,230-258,337)
    for k in ("norm_edge_00/gamma", "attention_gates_01/kernel", "dense_edge_b_00/bias", "norm_mha_01/beta",
              "dense_qkv_00/kernel", "dense_mha_01/bias", "dense_edge_r_00/kernel",
              "norm_fnn_node_00/gamma", "fnn_lr1_node_01/kernel", "fnn_lr2_edge_00/bias", "norm_fnn_edge_01/beta"):
        assert k in named, k
    assert len(named) == sum(1 for _ in st.parameters())
    assert tuple(named["dense_qkv_00/kernel"].shape) == (64, 192)        # Keras Dense kernels are [in, out]
    assert tuple(named["fnn_lr1_edge_00/kernel"].shape) == (64, 128)
    rng = np.random.default_rng(0)
    weights = {k: rng.standard_normal(tuple(p.shape)).astype(np.float32) for k, p in named.items()}
    missing, unexpected = st.load_keras_weights(weights)
    assert not missing and not unexpected
    for k, p in named.items():
        assert np.array_equal(p.detach().numpy(), weights[k])
    bad = dict(weights); bad.pop("dense_qkv_00/kernel")
    with pytest.raises(KeyError):
        st.load_keras_weights(bad)
    bad = dict(weights); bad["dense_qkv_00/kernel"] = np.zeros((192, 64), np.float32)
    with pytest.raises(ValueError):
        st.load_keras_weights(bad)


def test_ffn_constructor_errors_and_cpu_refusal():
    from egt_amd import FFN
    with pytest.raises(ValueError):
        FFN(64, ffn_multiplier=4.0)
    m = FFN(64)
    with pytest.raises(Exception):          # no CPU fallback: the HIP path refuses CPU tensors
        m(torch.zeros(4, 64))

def test_device_seeds_follow_next_seed_sequence():
    """egt_amd.graph.DeviceSeeds (CPU tensors here): word i walks the module's host-side next_seed() sequence."""
    import torch
    from egt_amd import EGT, DeviceSeeds
    a = [EGT(num_heads=8, random_mask_prob=0.1, seed=s) for s in (0, 5, 123456789)]
    b = [EGT(num_heads=8, random_mask_prob=0.1, seed=s) for s in (0, 5, 123456789)]
    for m in a + b:
        m._calls = 3
    seeds = DeviceSeeds(b, "cpu")
    for _ in range(4):
        seeds.advance()
        assert seeds.values() == [m.next_seed() for m in a]
    seeds.detach()
    assert [m._calls for m in b] == [m._calls for m in a] and all(m.seed_device is None for m in b)


def test_composed_operator_refuses_device_seed():
    import pytest
    import torch
    from egt_amd import EGT, DeviceSeeds
    m = EGT(num_heads=8, random_mask_prob=0.1, seed=1, edge_input=False, gate_input=False).train()
    DeviceSeeds([m], "cpu")
    with pytest.raises(RuntimeError, match="device-resident mask seeds"):
        m([torch.zeros(1, 4, 3 * 8 * 2)], mask=None)
