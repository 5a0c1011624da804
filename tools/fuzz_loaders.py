"""Mutation fuzz of the checkpoint parsers (safetensors, GGUF): random byte flips / truncations of valid files, every file that still
opens is converted tensor by tensor.  Meant to run on the AddressSanitizer build: tools/asan_check.sh."""
import sys, os, ctypes, tempfile, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import gguf_io as G
from mlimgsynth_amd import _lib
L=_lib.lib()
class TSEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("dtype", ctypes.c_int), ("n_dim", ctypes.c_int), ("shape", ctypes.c_int64 * 4), ("size", ctypes.c_size_t), ("data", ctypes.c_void_p)]
L.mlts_open.restype=ctypes.c_void_p; L.mlts_open.argtypes=[ctypes.c_char_p, ctypes.c_int]
L.mlts_close.argtypes=[ctypes.c_void_p]; L.mlts_count.argtypes=[ctypes.c_void_p]
L.mlts_at.restype=ctypes.POINTER(TSEntry); L.mlts_at.argtypes=[ctypes.c_void_p, ctypes.c_int]
L.mlts_entry_to_f32.argtypes=[ctypes.POINTER(TSEntry), ctypes.POINTER(ctypes.c_float), ctypes.c_int64]
d=tempfile.mkdtemp(); rng=np.random.default_rng(1)
tensors=[("a.f32", rng.standard_normal((3,5)).astype(np.float32), "F32"), ("b", rng.standard_normal((4,64)).astype(np.float32), "Q4_1"),
         ("model.diffusion_model.input_blocks.1.0.in_layers.2.weight", rng.standard_normal((8,4,3,3)).astype(np.float32), "F16"), ("c", rng.standard_normal((2,32)).astype(np.float32),"Q8_0")]
gg=os.path.join(d,"g.gguf"); G.write(gg, tensors, [("k", G.T_STR, "v"), ("arr", G.T_ARR, (G.T_STR, ["x","yy"]))])
from safetensors.numpy import save_file
st=os.path.join(d,"s.safetensors"); save_file({k: v.astype(np.float16 if kind=="F16" else np.float32) for k,v,kind in tensors}, st, metadata={"a":"b"})
n_open=0
for path in (gg, st):
    raw=open(path,'rb').read()
    hdr=min(len(raw), 600)
    for i in range(1500):
        b=bytearray(raw)
        for _ in range(rng.integers(1,4)):
            pos=rng.integers(0,hdr); b[pos]=rng.integers(0,256)
        if i%11==0: b=b[:rng.integers(4,len(b))]
        f=os.path.join(d,"f.bin"); open(f,'wb').write(b)
        for conv in (0,1):
            S=L.mlts_open(f.encode(), conv)
            if not S: continue
            n_open+=1
            for k in range(L.mlts_count(S)):
                e=L.mlts_at(S,k); n=1
                for j in range(4): n*=max(int(e.contents.shape[j]),0)
                if 0 < n < 1_000_000 and e.contents.dtype not in (26,27):
                    out=(ctypes.c_float*n)(); L.mlts_entry_to_f32(e, out, n)
            L.mlts_close(S)
# ---- LoRA adapters: mlts_open_lora + mlts_lora_apply on mutated adapter files (0-dimensional / zero-sized tensors, element
# counts that do not factor, integer tensors): every one must be REFUSED or applied, never crash
L.mlts_open_lora.restype=ctypes.c_void_p; L.mlts_open_lora.argtypes=[ctypes.c_char_p]
L.mlts_lora_apply.argtypes=[ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int]
base=os.path.join(d,"base.safetensors")
save_file({"model.diffusion_model.input_blocks.1.1.proj_in.weight": rng.standard_normal((16,8)).astype(np.float16)}, base)
lk="lora_unet_input_blocks_1_1_proj_in"
lora=os.path.join(d,"l.safetensors")
save_file({lk+".lora_down.weight": rng.standard_normal((4,8)).astype(np.float16), lk+".lora_up.weight": rng.standard_normal((16,4)).astype(np.float16),
           lk+".alpha": np.array(2.0, np.float32)}, lora)
raw=open(lora,'rb').read(); hdr=min(len(raw), 500); n_lora=0; n_app=0
for i in range(1500):
    b=bytearray(raw)
    for _ in range(rng.integers(1,4)):
        pos=rng.integers(8,hdr); b[pos]=rng.integers(32,127) if i%3 else rng.integers(0,256)
    if i%13==0: b=b[:rng.integers(16,len(b))]
    f=os.path.join(d,"fl.bin"); open(f,'wb').write(b)
    Ls=L.mlts_open_lora(f.encode())
    if not Ls: continue
    n_lora+=1
    D=L.mlts_open(base.encode(), 1)
    r=L.mlts_lora_apply(D, Ls, 0.5, 1)
    n_app+= r>0
    L.mlts_close(D); L.mlts_close(Ls)
# hand-made degenerate adapters: scalar down tensor, zero-sized inner dimension, counts that do not factor
for dn, up in (((), (16,4)), ((0,8), (16,0)), ((3,8), (16,4)), ((4,8), (5,5))):
    f=os.path.join(d,"deg.safetensors")
    save_file({lk+".lora_down.weight": np.zeros(dn, np.float16), lk+".lora_up.weight": np.zeros(up, np.float16)}, f)
    Ls=L.mlts_open_lora(f.encode())
    if Ls:
        D=L.mlts_open(base.encode(), 1)
        assert L.mlts_lora_apply(D, Ls, 1.0, 1) < 0, (dn, up)
        L.mlts_close(D); L.mlts_close(Ls)
print("fuzz ok; opened", n_open, "| lora files opened", n_lora, "applied", n_app)
