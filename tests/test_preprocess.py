"""Row a-1 (SigLipImageProcessor.preprocess, siglip_encoder.py:47-67): the integer restatement of Pillow's bicubic resize
(oracle/pil_bicubic.py) against Pillow itself and against the reference-generated fixture (CPU), and the HIP kernel behind
svln_preprocess_frames against all three, byte for byte (GPU)."""
import hashlib

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import pil_bicubic as PB
from oracle import streamvln_oracle as O
from streamvln_amd.config import TINY
from streamvln_amd.synthetic import synthetic_frame
from util import load_golden


def structured_frames():
    """same generators as oracle/make_golden.py: smooth ramps, a saturated checkerboard, thin stripes (bicubic overshoot -> clip8)"""
    y, x = np.mgrid[0:480, 0:640]
    grad = np.stack([(x * 255 // 639), (y * 255 // 479), ((x + y) * 255 // 1118)], -1).astype(np.uint8)
    checker = (((x // 3 + y // 5) % 2) * 255).astype(np.uint8)[..., None].repeat(3, -1)
    stripes = np.stack([((x % 7) < 2) * 255, ((y % 4) < 1) * 255, ((x * y) % 256)], -1).astype(np.uint8)
    return {"grad": grad, "checker": checker, "stripes": stripes}


def fixture_frames():
    frames = {f"synthetic_{s}": synthetic_frame(0, s) for s in range(4)}
    frames.update(structured_frames())
    return frames


ODD_SIZES = [(384, 384), (100, 37), (720, 1280), (384, 640), (480, 384), (97, 1111), (1, 1), (2, 2), (1080, 1920), (33, 3000)]


def pil_resize(frame, size=384):
    return np.asarray(Image.fromarray(frame).resize((size, size), Image.BICUBIC))


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def check_against_fixture(key, chw_f32, g):
    u8 = np.rint((chw_f32.astype(np.float64) * 0.5 + 0.5) * 255.0).astype(np.uint8)
    assert np.array_equal(u8[:, 100, :], g[f"{key}_row100_u8"]), key
    assert np.array_equal(u8.reshape(3, -1).astype(np.int64).sum(1), g[f"{key}_chan_sum_u8"]), key
    assert np.array_equal(sha(u8), g[f"{key}_sha256_u8"]), key
    assert np.array_equal(sha(chw_f32), g[f"{key}_sha256_f32"]), key


# ------------------------------------------------------------------------------------------------- CPU: the oracle is pinned
def test_restatement_equals_pillow_and_reference_fixture():
    g = load_golden("preprocess")
    for key, frame in fixture_frames().items():
        mine = PB.resize_bicubic_u8(frame, 384, 384)
        assert np.array_equal(mine, pil_resize(frame)), key
        full = PB.siglip_preprocess(frame)
        assert np.array_equal(full, O.siglip_preprocess(frame)), key
        check_against_fixture(key, full, g)                      # whole-frame hashes through the reference's own SigLipImageProcessor


@pytest.mark.parametrize("hw", ODD_SIZES)
def test_restatement_equals_pillow_on_odd_geometries(hw):
    rng = np.random.default_rng(hw[0] * 7919 + hw[1])
    frame = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    assert np.array_equal(PB.resize_bicubic_u8(frame, 384, 384), pil_resize(frame))
    assert np.array_equal(PB.resize_bicubic_u8(frame, 70, 70), pil_resize(frame, 70))       # other output sizes too


def test_coefficient_tables_are_what_pillow_uses():
    """640 -> 384: scale 5/3, support 10/3, 9 taps; rows sum to 2^22 within rounding; first/last windows are clipped"""
    ks, xmin, cnt, kk = PB.precompute_coeffs(640, 384)
    assert ks == 2 * 4 + 1 and xmin[0] == 0 and xmin[-1] + cnt[-1] == 640
    assert np.abs(kk.sum(1) - (1 << 22)).max() <= ks and kk.min() < 0            # negative lobes exist
    ks, xmin, cnt, kk = PB.precompute_coeffs(480, 384)
    assert ks == 2 * 3 + 1 and cnt.max() <= ks
    lut = PB.normalize_lut()
    assert lut[0] == -1.0 and lut[255] == 1.0 and np.all(np.diff(lut) > 0)


# ------------------------------------------------------------------------------------------------- GPU: the HIP kernel
@pytest.fixture(scope="module")
def model():
    from streamvln_amd.model import StreamVLNForCausalLM
    m = StreamVLNForCausalLM(TINY, dtype=torch.float32, max_envs=1, max_frames=1, max_positions=256)
    yield m
    m.close()


@pytest.mark.gpu
def test_hip_preprocess_is_bit_exact_vs_pillow_oracle_and_fixture(model):
    proc = model.get_vision_tower().image_processor
    assert proc.backend == "hip"
    g = load_golden("preprocess")
    for key, frame in fixture_frames().items():
        out = proc.preprocess_array(frame)
        assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (3, 384, 384)
        got = out.cpu().numpy()
        assert np.array_equal(got, O.siglip_preprocess(frame)), key              # Pillow on this box
        assert np.array_equal(got, PB.siglip_preprocess(frame)), key             # integer restatement
        check_against_fixture(key, got, g)                                        # reference-generated hashes


@pytest.mark.gpu
@pytest.mark.parametrize("hw", ODD_SIZES)
def test_hip_preprocess_odd_geometries(model, hw):
    rng = np.random.default_rng(hw[0] * 7919 + hw[1])
    frame = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    proc = model.get_vision_tower().image_processor
    got = proc.preprocess_array(frame).cpu().numpy()
    assert np.array_equal(got, O.siglip_preprocess(frame)), hw


@pytest.mark.gpu
def test_hip_preprocess_batch_pil_input_and_errors(model):
    from streamvln_amd import _lib
    proc = model.get_vision_tower().image_processor
    frames = [synthetic_frame(1, s) for s in range(5)]
    pv = proc.preprocess(images=[Image.fromarray(f) for f in frames], return_tensors="pt")["pixel_values"]     # one upload, one launch
    assert pv.is_cuda and tuple(pv.shape) == (5, 3, 384, 384)
    for f, got in zip(frames, pv.cpu().numpy()):
        assert np.array_equal(got, O.siglip_preprocess(f))
    one = proc.preprocess(images=Image.fromarray(frames[2]).convert("RGB"), return_tensors="pt")["pixel_values"][0]   # the reference call (streamvln_eval.py:271-274)
    assert np.array_equal(one.cpu().numpy(), O.siglip_preprocess(frames[2]))
    mixed = proc.preprocess(images=[frames[0], frames[1][:100, :200]], return_tensors="np")["pixel_values"]
    assert np.array_equal(mixed[1], O.siglip_preprocess(frames[1][:100, :200]))
    ms, n = model.preprocess_time(reset=True)
    assert n >= 8 and ms > 0
    with pytest.raises(ValueError):
        proc.preprocess_array(np.zeros((10, 10), dtype=np.uint8))
    with pytest.raises(_lib.SvlnError, match="100 x"):
        proc.preprocess_array(np.zeros((3000, 5, 3), dtype=np.uint8))             # Pillow >= 12 goes vertical-first there
    with pytest.raises(_lib.SvlnError, match="too large"):
        proc.preprocess_array(np.zeros((2160, 3840, 3), dtype=np.uint8))
    proc.backend = "pil"                                                          # explicit host path = Pillow itself
    assert np.array_equal(proc.preprocess_array(frames[0]).numpy(), O.siglip_preprocess(frames[0]))
    proc.backend = "hip"


@pytest.mark.gpu
def test_hip_preprocess_enqueue_ordering_and_blocking_entry(model):
    """The processor uses the enqueue-only entry (no host wait; ordered against torch's stream on the device): 300 back-to-back calls
    with distinct frames, results read only afterwards, must each be their own frame's output (double-buffered pinned staging, the
    timing-event ring wrapping around many times); the blocking C entry point gives the same bytes."""
    import ctypes as C
    from streamvln_amd import _lib
    proc = model.get_vision_tower().image_processor
    model.preprocess_time(reset=True)
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, (48, 64, 3), dtype=np.uint8) for _ in range(300)]
    outs = [proc.preprocess_array(f) for f in frames]                              # nothing synchronises in here
    sums = torch.stack([o.sum() for o in outs]).cpu().numpy()                      # torch work on the outputs: ordered behind the kernels
    for i in (0, 1, 2, 127, 255, 256, 257, 299):
        exp = O.siglip_preprocess(frames[i])
        assert np.array_equal(outs[i].cpu().numpy(), exp), i
        assert abs(float(sums[i]) - float(exp.astype(np.float64).sum())) < 1e-2 * max(1.0, abs(float(exp.sum())))
    ms, n = model.preprocess_time(reset=True)
    assert n == 300 and ms > 0
    out = torch.empty((1, 3, 384, 384), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    _lib.check(model._lib.svln_preprocess_frames(model._h, frames[7].ctypes.data_as(C.c_void_p), 1, 48, 64, 0, C.c_void_p(out.data_ptr())))
    assert np.array_equal(out[0].cpu().numpy(), O.siglip_preprocess(frames[7]))    # complete on return: no sync before the read-back needed


@pytest.mark.gpu
def test_closed_model_processor_raises():
    from streamvln_amd.model import StreamVLNForCausalLM
    m = StreamVLNForCausalLM(TINY, dtype=torch.float32, max_envs=1, max_frames=1, max_positions=256)
    proc = m.get_vision_tower().image_processor
    proc.preprocess_array(np.zeros((48, 64, 3), dtype=np.uint8))
    m.close()
    with pytest.raises(RuntimeError, match="closed"):
        proc.preprocess_array(np.zeros((48, 64, 3), dtype=np.uint8))


@pytest.mark.gpu
def test_frame_ring_is_read_in_place_and_bit_exact():
    """svln_frame_ring: frames written into the engine's pinned ring and handed over as slot views are read by the GPU where they are
    (no staging copy) and give the bytes of the staging path / of Pillow; padded slots (frame bytes not a multiple of 256), a
    multi-frame batch straight from consecutive slots, a pointer inside the ring that is not 16-byte aligned (falls back to staging),
    slot reuse after svln_frame_ring_wait, and replacing the ring."""
    import ctypes as C
    from streamvln_amd import _lib
    from streamvln_amd.model import StreamVLNForCausalLM
    m = StreamVLNForCausalLM(TINY, dtype=torch.float32, max_envs=1, max_frames=1, max_positions=256)
    proc = m.get_vision_tower().image_processor
    ring = m.frame_ring(6, 480, 640)
    assert ring.shape == (6, 480, 640, 3) and ring.dtype == np.uint8 and ring.strides[0] == 480 * 640 * 3       # 921 600 = 3600 * 256
    frames = [synthetic_frame(2, s) for s in range(6)]
    for k in range(6):
        ring[k][...] = frames[k]
    outs = [proc.preprocess_array(ring[k]) for k in range(6)]                     # enqueue-only calls, nothing synchronises in between
    for k in (0, 3, 5):
        assert np.array_equal(outs[k].cpu().numpy(), O.siglip_preprocess(frames[k])), k
    out = torch.empty((3, 3, 384, 384), dtype=torch.float32, device="cuda")       # three consecutive slots as ONE batch, straight from the ring
    torch.cuda.synchronize()
    _lib.check(m._lib.svln_preprocess_frames(m._h, ring[2].ctypes.data, 3, 480, 640, 0, out.data_ptr()))
    for j in range(3):
        assert np.array_equal(out[j].cpu().numpy(), O.siglip_preprocess(frames[2 + j])), j
    sub = ring[1].reshape(-1)[3:3 + 48 * 64 * 3].reshape(48, 64, 3)               # inside the ring but 3 bytes off a 16-byte boundary
    assert np.array_equal(proc.preprocess_array(sub).cpu().numpy(), O.siglip_preprocess(np.array(sub)))
    m.frame_ring_wait(0)                                                           # the upload that read slot 0 has run: the slot may be rewritten
    ring[0][...] = frames[5]
    assert np.array_equal(proc.preprocess_array(ring[0]).cpu().numpy(), O.siglip_preprocess(frames[5]))
    with pytest.raises(_lib.SvlnError, match="no such slot"):
        m.frame_ring_wait(6)
    # a geometry whose frame size is not a multiple of 256 bytes: padded slots
    ring2 = m.frame_ring(3, 97, 1111)
    assert ring2.strides[0] % 256 == 0 and ring2.strides[0] >= 97 * 1111 * 3
    rng = np.random.default_rng(11)
    odd = [rng.integers(0, 256, (97, 1111, 3), dtype=np.uint8) for _ in range(3)]
    for k in range(3):
        ring2[k][...] = odd[k]
    for k in range(3):
        assert np.array_equal(proc.preprocess_array(ring2[k]).cpu().numpy(), O.siglip_preprocess(odd[k])), k
    pv = proc.preprocess(images=[ring2[0], ring2[1]], return_tensors="pt")["pixel_values"]     # (a stacked copy: staging path)
    assert np.array_equal(pv[1].cpu().numpy(), O.siglip_preprocess(odd[1]))
    m.close()
