"""Mutation fuzz of the command line's own file readers (csrc/cli/main.c: PNG incl. inflate, PNM): `convert -i <file> -o <out>` on corrupted
images must fail with a message or succeed, never crash (run on the AddressSanitizer build of the CLI: tools/asan_check.sh).
usage: python3 tools/fuzz_cli_readers.py [cli binary] [iterations]"""
import os, subprocess, sys, tempfile, zlib, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cli = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mlimgsynth_amd", "bin", "mlimgsynth-amd")
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(3)
d = tempfile.mkdtemp()


def png(w, h, c, level):
    raw = b"".join(b"\x00" + rng.integers(0, 256, w * c, dtype=np.uint8).tobytes() for _ in range(h))
    def chunk(t, b): return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    ct = {1: 0, 3: 2, 4: 6}[c]
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ct, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, level)) + chunk(b"IEND", b"")


seeds = [png(17, 9, 3, 9), png(8, 8, 4, 1), png(5, 30, 1, 6), b"P6\n7 5\n255\n" + bytes(range(105)), b"P5\n# c\n4 4\n255\n" + bytes(16)]
# hostile headers: dimensions whose products wrap 32 / 64 bits, endless digits
seeds += [b"P6\n4294967295 4294967295\n255\n" + bytes(64), b"P5\n" + b"9" * 40 + b" 2\n255\n" + bytes(64),
          png(17, 9, 3, 9).replace(struct.pack(">II", 17, 9), struct.pack(">II", 0xFFFFFFFF, 0xFFFFFFFE))]
bad = 0
for i in range(iters):
    b = bytearray(seeds[i % len(seeds)])
    if i >= len(seeds):
        for _ in range(rng.integers(1, 5)):
            b[rng.integers(0, len(b))] = rng.integers(0, 256)
        if i % 7 == 0: b = b[:rng.integers(1, len(b))]
    f = os.path.join(d, "in.bin"); open(f, "wb").write(bytes(b))
    r = subprocess.run([cli, "convert", "-i", f, "-o", os.path.join(d, "out.pnm")], capture_output=True, timeout=60)
    if r.returncode not in (0, 1):
        bad += 1
        print("iteration", i, "exit code", r.returncode, r.stderr[-300:])
print("cli reader fuzz:", iters, "files,", bad, "abnormal exits")
sys.exit(1 if bad else 0)
