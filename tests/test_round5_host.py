"""Host-side behaviour fixed in round 5 (ADVICE.md of round 4); no GPU needed."""
import warnings

import numpy as np
import pytest
import torch


def _png(path, arr, mode):
    from PIL import Image
    Image.fromarray(arr, mode).save(path)


def test_map_assigned_after_a_recorded_tile_is_not_tiled_again(monkeypatch):
    """base.py:524-537 repeats the maps that ARE there; a map assigned afterwards keeps its size.  A recorded (lazy) tile must
    therefore be carried out before the assignment lands -- ADVICE r4 (medium): roughness came back 32x32 beside a 16x16 albedo."""
    from PIL import Image
    import pypbr_amd.materials as M
    monkeypatch.setattr(M, "DEFER_IMAGE_DECODE", True)
    rng = np.random.default_rng(3)
    rgb = (rng.random((8, 8, 3)) * 255).astype(np.uint8)
    grey = (rng.random((8, 8)) * 255).astype(np.uint8)
    m = M.BasecolorMetallicMaterial(albedo=Image.fromarray(rgb, "RGB"), roughness=Image.fromarray(grey, "L"))
    m.tile(2)
    assert m.lazy_tile == (2, 2)                       # maps nobody has seen: the repeat is only recorded
    m.roughness = torch.full((1, 16, 16), 0.5)
    assert m.lazy_tile == (1, 1)
    maps = m._maps
    assert maps["albedo"].shape == (3, 16, 16) and maps["roughness"].shape == (1, 16, 16)
    assert torch.equal(maps["roughness"], torch.full((1, 16, 16), 0.5))
    assert torch.equal(maps["albedo"], (torch.from_numpy(rgb.transpose(2, 0, 1).copy()).float() / 255).repeat(1, 2, 2))
    # an explicit lazy tile on plain float maps behaves the same
    p = M.BasecolorMetallicMaterial(albedo=torch.rand(3, 4, 4), roughness=torch.rand(1, 4, 4))
    p.tile(3, lazy=True)
    p.metallic = torch.zeros(1, 12, 12)
    assert p.size == (12, 12) and {k: tuple(v.shape[-2:]) for k, v in p._maps.items()} == {"albedo": (12, 12), "roughness": (12, 12), "metallic": (12, 12)}


def test_loader_closes_every_file_it_opened(tmp_path, monkeypatch):
    """the image of the workflow NOT chosen (popped by select_material_class) and every image after a decoder raised
    stayed open until garbage collection."""
    import pypbr_amd.io as IO
    from PIL import Image
    rng = np.random.default_rng(5)
    for name, mode in (("basecolor", "RGB"), ("diffuse", "RGB"), ("roughness", "L"), ("metallic", "L"), ("specular", "RGB")):
        shape = (6, 10, 3) if mode == "RGB" else (6, 10)
        _png(tmp_path / (name + ".png"), (rng.random(shape) * 255).astype(np.uint8), mode)
    opened = []
    real_open = Image.open
    monkeypatch.setattr(Image, "open", lambda *a, **k: (opened.append(real_open(*a, **k)), opened[-1])[1])

    def closed(im):
        return getattr(im, "fp", None) is None

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        IO.load_material_from_folder(str(tmp_path), preferred_workflow="metallic")
    assert len(opened) == 5 and all(closed(im) for im in opened)
    opened.clear()
    monkeypatch.setattr(IO, "_decoded", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("decoder failed")))
    with pytest.raises(RuntimeError), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        IO.load_material_from_folder(str(tmp_path), preferred_workflow="metallic")
    assert len(opened) == 5 and all(closed(im) for im in opened)


def test_upload_stage_is_released_on_request_and_staging_refuses_device_sources():
    """(low) the per-thread page-locked staging block can be dropped; (high) the host memcpy of the staging copy is only
    ever taken for a host source."""
    import inspect
    from pypbr_amd import functional as F
    buf, slot = F._upload_stage(1024)
    assert buf.numel() == 1024 and slot is not None and getattr(F._UPLOAD_STAGE, "slot", None) is slot
    F.release_upload_stage()
    assert getattr(F._UPLOAD_STAGE, "slot", None) is None
    F.release_upload_stage()                           # idempotent
    dst = torch.zeros(64, dtype=torch.uint8)
    src = torch.arange(16, dtype=torch.float32)
    F._stage_copy(dst, src, 64)
    assert torch.equal(dst.view(torch.float32), src)
    assert 'src.device.type == "cpu"' in inspect.getsource(F._stage_copy)


def test_resident_sends_only_host_maps_through_the_packed_upload(monkeypatch):
    """material.to('cuda:1') of maps on cuda:0 went through upload_packed's host memcpy.  Without a second GPU the
    routing is checked on the decision itself: maps whose device type is not 'cpu' take Tensor.to(compute)."""
    import pypbr_amd.materials as M
    from pypbr_amd import functional as F

    class FakeDev:
        """Stands in for a tensor on another GPU: only what _resident looks at."""
        def __init__(self, t, type_):
            self.t, self._type = t, type_
            self.requires_grad, self.dtype, self.shape = False, t.dtype, t.shape
            self.moved = False

        @property
        def device(self):
            return type("D", (), {"type": self._type, "__eq__": lambda s, o: False, "__ne__": lambda s, o: True, "__hash__": lambda s: 0})()

        def to(self, dev):
            self.moved = True
            return self.t

        def dim(self):
            return self.t.dim()

    m = M.BasecolorMetallicMaterial()
    other = FakeDev(torch.rand(3, 4, 4), "cuda")
    m._raw["albedo"] = other
    host = torch.rand(1, 4, 4)
    m._raw["roughness"] = host
    sent = []
    monkeypatch.setattr(M, "_compute_device", lambda home: torch.device("cpu"))           # no GPU here: "compute" is a stand-in
    monkeypatch.setattr(F, "upload_packed", lambda ts, dev, **k: (sent.extend(ts), ([t for t in ts], None))[1])
    # roughness sits on "compute" (cpu == cpu) already; albedo is away on another device of type cuda
    out = m._resident(keep=False)
    assert other.moved and out["albedo"] is other.t and sent == []
