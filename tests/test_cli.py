"""mlimgsynth-amd, the command line front end (counterpart of src/main_mlimgsynth.c): option plumbing and image / tensor file
I/O on CPU, a generation on the GPU."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "mlimgsynth_amd", "bin", "mlimgsynth-amd")


def run(*args, ok=True):
    r = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=600)
    if ok:
        assert r.returncode == 0, r.stderr
    return r


def png_chunks(b):
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    p, out = 8, []
    while p < len(b):
        n = struct.unpack(">I", b[p:p + 4])[0]
        typ, data, crc = b[p + 4:p + 8], b[p + 8:p + 8 + n], struct.unpack(">I", b[p + 8 + n:p + 12 + n])[0]
        assert zlib.crc32(typ + data) == crc
        out.append((typ, data))
        p += 12 + n
    return out


def png_pixels(b):
    ch = png_chunks(b)
    w, h, bd, ct = struct.unpack(">IIBB", ch[0][1][:10])
    c = {0: 1, 2: 3, 6: 4}[ct]
    raw = zlib.decompress(b"".join(d for t, d in ch if t == b"IDAT"))
    rows = np.frombuffer(raw, np.uint8).reshape(h, w * c + 1)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, c), dict((d.split(b"\0", 1)[0], d.split(b"\0", 1)[1]) for t, d in ch if t == b"tEXt")


def write_png_filtered(path, img):
    """8-bit PNG with zlib level 9 (dynamic Huffman blocks) and a different filter type on every row (all five kinds)"""
    h, w, c = img.shape
    raw = bytearray()
    prev = np.zeros(w * c, np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        ft = y % 5
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        ul = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - (left + prev) // 2
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            f = cur - pred
        raw.append(ft)
        raw += (f & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    z = zlib.compress(bytes(raw), 9)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, {1: 0, 3: 2, 4: 6}[c], 0, 0, 0)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("c", [1, 3, 4])
def test_image_files_round_trip(tmp_path, c):
    rng = np.random.default_rng(c)
    img = (rng.integers(0, 256, (37, 53, c)) // 8 * 8).astype(np.uint8)          # compressible: exercises back-references
    img[5:20, 7:30] = 200
    src = str(tmp_path / "in.png")
    write_png_filtered(src, img)
    out_png = str(tmp_path / "out.png")
    run("convert", "-i", src, "-o", out_png)
    px, _ = png_pixels(open(out_png, "rb").read())
    assert np.array_equal(px, img)
    if c != 4:
        out_pnm = str(tmp_path / ("out.ppm" if c == 3 else "out.pgm"))
        run("convert", "-i", out_png, "-o", out_pnm)
        b = open(out_pnm, "rb").read()
        hdr = b"P%c\n53 37\n255\n" % (b"6" if c == 3 else b"5")
        assert b.startswith(hdr) and np.array_equal(np.frombuffer(b[len(hdr):], np.uint8).reshape(37, 53, c), img)
        run("convert", "-i", out_pnm, "-o", str(tmp_path / "again.png"))
        assert np.array_equal(png_pixels(open(tmp_path / "again.png", "rb").read())[0], img)


def test_options_travel_through_the_library_grammar(tmp_path):
    assert "generate" in run("--help").stdout and "0.4.2" in run("-V").stdout
    r = run("generate", "--steps", "12x", ok=False)
    assert r.returncode == 1 and "invalid argument '12x' for option 'steps'" in r.stderr        # the library's message, not the CLI's
    assert run("generate", "-x", ok=False).returncode == 1
    assert run("frobnicate", ok=False).returncode == 1
    assert "not implemented" in run("check", ok=False).stderr
    assert run("list-backends").stdout.startswith("HIP")
    r = run("generate", "-m", str(tmp_path / "missing.safetensors"), "-p", "x", ok=False)
    assert r.returncode == 1 and ("missing.safetensors" in r.stderr or "no CPU fallback" in r.stderr)   # on a machine without a GPU the set-up fails first, loudly


@pytest.mark.gpu
def test_generate_vae_and_tensor_files_on_gpu(tmp_path):
    """the CLI drives a complete generation of the synthetic tiny model: PNG with the infotext in a tEXt chunk, --olatent in
    the reference's TENSOR format, img2img from that PNG, vae-decode of the saved latent = the generated image"""
    out, lat = str(tmp_path / "o.png"), str(tmp_path / "o.tensor")
    base = ["-m", "synth:tiny", "-d", "64,64", "-s", "4", "--method", "euler_a", "-S", "42", "--cfg-scale", "7",
            "--tokens", "5,17,300,42,7", "--ntokens", "9,9,8"]
    r = run("generate", *base, "-o", out, "--olatent", lat)
    assert "Denoising 4/4" in r.stderr and "Saved" in r.stderr
    px, text = png_pixels(open(out, "rb").read())
    assert px.shape == (64, 64, 3) and b"Steps: 4" in text[b"parameters"] and b"Seed: 42" in text[b"parameters"]
    b = open(lat, "rb").read()
    assert b.startswith(b"TENSOR F32 8 8 4 1\n") and len(b) == 19 + 8 * 8 * 4 * 4
    # the same generation through the Python class gives the same pixels
    from mlimgsynth_amd import mlimgsynth as W
    import ctypes as C
    with W.MLImgSynth() as w:
        for k, v in (("model", "synth:tiny"), ("image-dim", "64,64"), ("steps", 4), ("method", "euler_a"), ("seed", 42), ("cfg-scale", 7)):
            w.option_set(k, v)
        f = w._lib.mlis_amd_prompt_tokens_set
        f.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int, C.c_int]
        for ids, neg in (([5, 17, 300, 42, 7], 0), ([9, 9, 8], 1)):
            a = np.array(ids, np.int32)
            assert f(w._ctx, a.ctypes.data_as(C.POINTER(C.c_int32)), None, a.size, neg) > 0
        w.generate()
        assert np.array_equal(w.image_get(0).numpy(), px)
    # vae-decode of the saved latent reproduces the image; batch of 2 writes two numbered files; img2img runs from the PNG
    run("vae-decode", "-m", "synth:tiny", "--ilatent", lat, "-o", str(tmp_path / "d.png"))
    assert np.array_equal(png_pixels(open(tmp_path / "d.png", "rb").read())[0], px)
    run("generate", *base, "--batch-size", "2", "-o", str(tmp_path / "b.png"))
    b1, b2 = png_pixels(open(tmp_path / "b-1.png", "rb").read())[0], png_pixels(open(tmp_path / "b-2.png", "rb").read())[0]
    assert np.array_equal(b1, px) and not np.array_equal(b2, px)
    run("generate", *base, "-i", out, "--f-t-ini", "0.5", "-o", str(tmp_path / "i.png"))
    i2 = png_pixels(open(tmp_path / "i.png", "rb").read())[0]
    assert i2.shape == (64, 64, 3) and not np.array_equal(i2, px)
    run("vae-test", "-m", "synth:tiny", "-i", out, "-o", str(tmp_path / "t.png"))
    assert png_pixels(open(tmp_path / "t.png", "rb").read())[0].shape == (64, 64, 3)
