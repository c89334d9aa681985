"""Shared helpers for the test-suite (skeletons, tolerances)."""
PARENTS = {
    17: [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15],            # reference reconstruction.py:95
    19: [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 10, 13, 14, 10, 16, 17],  # reference reconstruction.py:87
    15: [-1, 0, 1, 2, 3, 1, 5, 6, 0, 8, 9, 0, 11, 12, 1],                    # reference common/humaneva_dataset.py:7
    16: [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 8, 10, 11, 8, 13, 14],               # reference h36m_dataset.py:267-277
}


def perturb_like_golden(model, gen):
    """The deterministic perturbation of tests/golden/make_golden.py::perturb: BatchNorm affine parameters and running statistics,
    C_k, e and the attention biases moved off their trivial initial values, drawn from `gen` in named_parameters() /
    named_buffers() order.  Works on the reference's modules and on the drop-in's (same names, same order)."""
    import torch
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('_bn.weight') or '.bn_1.weight' in name or '.bn_2.weight' in name or \
                    name.startswith('layers_bn.') and name.endswith('weight') or name in ('init_bn.weight', 'expand_bn.weight'):
                p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
            elif name.endswith('bn.bias') or '.bn_1.bias' in name or '.bn_2.bias' in name or \
                    name.startswith('layers_bn.') and name.endswith('bias') or name in ('init_bn.bias', 'expand_bn.bias'):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
            elif name.endswith('C_k'):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
            elif name.endswith('.e'):
                p.copy_(1.0 + torch.randn(p.shape, generator=gen) * 0.3)
            elif name.endswith('.bias'):  # g/theta/phi conv1d biases
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
        for name, b in model.named_buffers():
            if name.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
            elif name.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)


def state_digest(state_dict):
    """SHA-256 over (key, raw bytes) of a state_dict: the seed-constructed fixtures store it, the tests assert it on the drop-in's
    seed-constructed model before comparing outputs (a mismatch means "different weights", not "wrong kernel")."""
    import hashlib
    h = hashlib.sha256()
    for k, v in state_dict.items():
        h.update(k.encode())
        h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


# Inference fixtures (tests/golden/make_golden_inference.py): the configurations the reference's demo scripts build.
INFERENCE_CASES = {
    # gen_skes.py:43-69 / tools/inference.py:73-91: causal single-frame-batching models
    'baseball_causal27': dict(cls='strided', arc=[3, 3, 3], channels=128, causal=True, seed=2701),
    'baseball_causal81': dict(cls='strided', arc=[3, 3, 3, 3], channels=64, causal=True, seed=8101),
    # reconstruction.py:219-258: symmetric models over the whole edge-padded clip
    'baseball_sym27': dict(cls='dilated', arc=[3, 3, 3], channels=128, causal=False, seed=2702),
    'baseball_sym243': dict(cls='dilated', arc=[3, 3, 3, 3, 3], channels=32, causal=False, seed=24301),
    # the causal DILATED model over the clip (reconstruction.py --causal): what gast_hip.streaming reproduces frame by frame
    'baseball_causal27_dil': dict(cls='dilated', arc=[3, 3, 3], channels=128, causal=True, seed=2703),
}
SHAPE243 = dict(arc=[3, 3, 3, 3, 3], channels=32, B=2, T=245, seed=24302)       # reconstruction.py:225-227
