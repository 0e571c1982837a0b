"""View / light / intensity tensors that live on the device (ABI 5: pbr_render_desc.device_params, pbr_prepare_device_params): the kernels
read them from device memory instead of the kernel-argument segment.  The reference runs plain torch ops on such tensors
(cooktorrance.py:95-96, :126-140), so a light that an optimiser fits to a photograph never leaves the device; here too now -- same
images, same gradients as with host parameters, no blocking read-back, and the whole fitting step captures into a HIP graph whose replays
see the light's current value."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["operator", "ctypes"], autouse=True)
def binding(request):
    """Both bindings of the C ABI: torch.ops.pbr_hip.* (csrc/torch_ops.cpp) and the ctypes plan."""
    from pypbr_amd import functional as F
    saved = F.USE_TORCH_OPS
    F.USE_TORCH_OPS = request.param == "operator"
    yield request.param
    F.USE_TORCH_OPS = saved


def _maps(B, H, W, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(B, 3, H, W, generator=g)
    n = torch.nn.functional.normalize(torch.cat([torch.rand(B, 2, H, W, generator=g) - 0.5, torch.ones(B, 1, H, W)], 1), dim=1)
    r = torch.rand(B, 1, H, W, generator=g) * 0.7 + 0.25
    m = torch.rand(B, 1, H, W, generator=g)
    return [t.cuda().to(dtype) for t in (a, n, r, m)]


@pytest.mark.parametrize("light_type", ["point", "directional"])
@pytest.mark.parametrize("lights", [1, 3])
@pytest.mark.parametrize("dtype,B", [(torch.float32, 1), (torch.float16, 4)])
def test_device_parameters_give_the_host_parameters_image_bit_for_bit(light_type, lights, dtype, B):
    """The folding (normalise, V + L, 1/|V + L|^2, the Fresnel power) is ONE function compiled for host and device with fp contraction off
    (ct_launch.hpp: fold_light): the device block holds the bits the host would have put into the kernel arguments."""
    from pypbr_amd import functional as F
    maps = _maps(B, 40, 72, 5, dtype)
    view = [0.1, -0.2, 1.0]
    L = [[0.3, 0.2, 1.1], [-0.4, 0.1, 0.8], [0.0, -0.5, 0.9]][:lights]
    I = [[1.0, 0.9, 0.8], [0.5, 0.5, 0.5], [0.2, 0.7, 0.4]][:lights]
    kw = dict(light_type=light_type, light_size=1.5)
    host = F.cook_torrance(*maps, view_dir=view, light=L if lights > 1 else L[0], light_intensity=I if lights > 1 else I[0], **kw)
    dev = F.cook_torrance(*maps, view_dir=torch.tensor(view).cuda(), light=torch.tensor(L if lights > 1 else L[0]).cuda(),
                          light_intensity=torch.tensor(I if lights > 1 else I[0]).cuda(), **kw)
    assert torch.equal(host, dev)
    mixed = F.cook_torrance(*maps, view_dir=view, light=torch.tensor(L if lights > 1 else L[0]).cuda(), light_intensity=I if lights > 1 else I[0], **kw)
    assert torch.equal(host, mixed)                               # one device tensor is enough: the others are copied up, not down
    one = F.cook_torrance(*maps, view_dir=view, light=torch.tensor(L).cuda(), light_intensity=torch.tensor([0.6, 0.6, 0.6]).cuda(), **kw)
    assert torch.equal(one, F.cook_torrance(*maps, view_dir=view, light=L, light_intensity=[0.6, 0.6, 0.6], **kw))      # one (grey) intensity for all


def test_no_host_read_of_device_parameters():
    """`.cpu()` / `.tolist()` of the parameter tensors would synchronise; under torch.cuda.graph capture it raises.  The evaluation, its
    backward (maps AND parameters) and the fused loss step all capture."""
    from pypbr_amd import functional as F
    maps = _maps(1, 32, 64, 7)
    view, light, inten = torch.tensor([0.0, 0.1, 1.0]).cuda(), torch.tensor([0.2, 0.1, 1.0]).cuda(), torch.tensor([1.0, 0.9, 0.8]).cuda()
    F.cook_torrance(*maps, view_dir=view, light=light, light_intensity=inten)            # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = F.cook_torrance(*maps, view_dir=view, light=light, light_intensity=inten)
    graph.replay()
    first = out.clone()
    light.copy_(torch.tensor([-0.3, 0.3, 0.8]))                                            # the graph reads the tensor at replay time
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(first, F.cook_torrance(*maps, view_dir=[0.0, 0.1, 1.0], light=[0.2, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8]))
    assert torch.equal(out, F.cook_torrance(*maps, view_dir=[0.0, 0.1, 1.0], light=[-0.3, 0.3, 0.8], light_intensity=[1.0, 0.9, 0.8]))


@pytest.mark.parametrize("light_type", ["point", "directional"])
def test_gradients_with_device_parameters_equal_those_with_host_parameters(light_type):
    from pypbr_amd import functional as F
    maps = _maps(2, 24, 46, 11)
    wt = torch.rand(2, 3, 24, 46, generator=torch.Generator().manual_seed(12)).cuda() - 0.4
    vals = ([0.05, -0.1, 1.0], [[0.3, 0.2, 1.1], [-0.4, 0.1, 0.8]], [[1.0, 0.9, 0.8], [0.4, 0.5, 0.6]])
    got = {}
    for where in ("cpu", "cuda"):
        leaves = [t.clone().requires_grad_(True) for t in maps]
        params = [torch.tensor(v, device=where, requires_grad=True) for v in vals]
        out = F.cook_torrance(*leaves, view_dir=params[0], light=params[1], light_intensity=params[2], light_type=light_type, light_size=2.0)
        (out * wt).sum().backward()
        assert all(p.grad is not None and p.grad.device.type == where for p in params)
        got[where] = [t.grad for t in leaves] + [p.grad.cpu() for p in params]
    for x, y in zip(got["cpu"], got["cuda"]):
        assert torch.equal(x.cpu(), y.cpu())


def test_a_light_fitting_step_captured_into_a_hip_graph():
    """Fit a point light's position to a target image: forward (parameters on the device), MSE, backward to the light (the light-gradient
    kernels), SGD update of the light tensor -- one captured graph, replayed; the same steps run eagerly with a host-resident light
    reach the same position (to fp32 rounding of the optimiser arithmetic), and the loss falls."""
    from pypbr_amd import functional as F
    maps = _maps(1, 48, 64, 21)
    kw = dict(view_dir=[0.0, 0.0, 1.0], light_intensity=[1.0, 1.0, 1.0], light_type="point", light_size=1.0)
    target = F.cook_torrance(*maps, light=[0.25, -0.15, 0.9], **kw)
    start = [-0.2, 0.2, 1.2]
    lr = 0.5

    def step(light):
        light.grad = None
        loss = torch.nn.functional.mse_loss(F.cook_torrance(*maps, light=light, **kw), target)
        loss.backward()
        with torch.no_grad():
            light.add_(light.grad, alpha=-lr)
        return loss

    dev_light = torch.tensor(start, device="cuda", requires_grad=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(dev_light)
    torch.cuda.current_stream().wait_stream(side)
    with torch.no_grad():
        dev_light.copy_(torch.tensor(start))
    dev_light.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step(dev_light)
    losses = []
    for _ in range(60):
        graph.replay()
        losses.append(float(loss.detach()))
    host_light = torch.tensor(start, requires_grad=True)
    for _ in range(60):
        host_loss = step(host_light)
    assert losses[-1] < 0.6 * losses[0]
    assert torch.allclose(dev_light.detach().cpu(), host_light.detach(), rtol=0, atol=2e-5), (dev_light, host_light)
    assert abs(losses[-1] - float(host_loss.detach())) <= 1e-6
