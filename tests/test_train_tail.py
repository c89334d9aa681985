"""Training-step tail: fused mpjpe and the flat Adam/AMSGrad step."""
import numpy as np
import pytest
import torch


def _mlp():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))


def test_flat_adam_host_logic_cpu():
    """Re-homing keeps values, adopts the FlatGradAllReduce buffer, refuses to step on CPU, exports torch-format state."""
    from gast_hip.optim import FlatAdam
    from gast_hip.dist import FlatGradAllReduce
    m = _mlp()
    before = [p.detach().clone() for p in m.parameters()]
    sync = FlatGradAllReduce(m.parameters())
    opt = FlatAdam(m.parameters(), lr=1e-3, amsgrad=True)
    for a, p in zip(before, m.parameters()):
        assert torch.equal(a, p.detach())
    st = opt._flat[0]
    off = 0
    for p in m.parameters():                       # every parameter is a view of the flat buffer, tightly packed in order
        assert p.data_ptr() == st['P'].data_ptr() + 4 * off
        off += p.numel()
    assert opt.flat_grads()[0].data_ptr() == sync.flat.data_ptr()
    m(torch.randn(4, 5)).sum().backward()
    assert m[0].weight.grad.data_ptr() == sync.flat.data_ptr()
    assert float(sync.flat.abs().sum()) > 0
    opt.zero_grad()
    assert float(sync.flat.abs().sum()) == 0
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        opt.step()
    sd = opt.state_dict()
    ref = torch.optim.Adam(_mlp().parameters(), lr=1e-3, amsgrad=True).state_dict()
    assert sd['param_groups'][0]['params'] == ref['param_groups'][0]['params']
    assert set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'}
    with pytest.raises(ValueError):
        FlatAdam(_mlp().parameters(), lr=-1.0)


def test_mpjpe_refuses_cpu():
    from gast_hip.loss import mpjpe
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        mpjpe(torch.zeros(2, 1, 17, 3), torch.zeros(2, 1, 17, 3))


@pytest.mark.gpu
@pytest.mark.parametrize('amsgrad,wd', [(True, 0.0), (False, 0.0), (True, 0.01)])
def test_flat_adam_matches_torch(amsgrad, wd):
    """Ten steps of the flat HIP Adam against torch.optim.Adam on the same gradients (reference trainval.py:78)."""
    from gast_hip.optim import FlatAdam
    ma, mb = _mlp().cuda(), _mlp().cuda()
    # an odd-sized extra parameter exercises the scalar tail and unaligned views
    ma.extra = torch.nn.Parameter(torch.randn(5, device='cuda'))
    mb.extra = torch.nn.Parameter(ma.extra.detach().clone())
    oa = FlatAdam(ma.parameters(), lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, amsgrad=amsgrad)
    ob = torch.optim.Adam(mb.parameters(), lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, amsgrad=amsgrad)
    gen = torch.Generator().manual_seed(0)
    for it in range(10):
        x = torch.randn(16, 5, generator=gen).cuda()
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad()
            (m(x).pow(2).mean() + m.extra.pow(2).sum() * (0.1 if it % 3 else 2.0)).backward()
            o.step()
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    sa, sb = oa.state_dict(), ob.state_dict()
    for i in sb['state']:
        for k in sb['state'][i]:
            np.testing.assert_allclose(sa['state'][i][k].cpu().numpy(), sb['state'][i][k].cpu().numpy(), rtol=3e-5, atol=1e-9, err_msg=k)
    # state round trip
    mc = _mlp().cuda()
    mc.extra = torch.nn.Parameter(torch.zeros(5, device='cuda'))
    oc = FlatAdam(mc.parameters(), lr=3e-3, amsgrad=amsgrad)
    oc.load_state_dict(sa)
    sc = oc.state_dict()
    for i in range(5):
        for k in sc['state'][i]:
            assert torch.equal(sc['state'][i][k].cpu(), sa['state'][i][k].cpu()), k


@pytest.mark.gpu
@pytest.mark.parametrize('shape,tshape', [((128, 1, 17, 3), (128, 1, 17, 3)), ((4, 5, 17, 3), (4, 1, 17, 3)), ((3, 2, 15, 2), (3, 2, 15, 2))])
def test_mpjpe_matches_reference_formula(shape, tshape):
    """reference common/loss.py:5-11: torch.mean(torch.norm(predicted - target, dim=-1)); value and gradient, 1e-6."""
    from gast_hip.loss import mpjpe
    gen = torch.Generator().manual_seed(1)
    pred = torch.randn(*shape, generator=gen).cuda().requires_grad_(True)
    tgt = torch.randn(*tshape, generator=gen).cuda()
    with torch.no_grad():
        pred[0, 0, 0] = tgt[0, 0, 0]           # a zero-length error vector: gradient 0, not NaN
    loss = mpjpe(pred, tgt)
    (loss * 1.7).backward()
    p2 = pred.detach().clone().requires_grad_(True)
    ref = torch.mean(torch.norm(p2 - tgt, dim=-1))
    (ref * 1.7).backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    g, gr = pred.grad, torch.nan_to_num(p2.grad, nan=0.0)
    assert torch.isfinite(g).all()
    np.testing.assert_allclose(g.cpu().numpy(), gr.cpu().numpy(), rtol=1e-5, atol=1e-8)
