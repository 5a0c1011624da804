"""Timeline of one weight-streaming UNet evaluation (MLSD_WSTREAM_TRACE=1: per segment, upload and compute intervals on the device).
usage: python3 tools/wstream_trace.py [model] [latent] [N] [slab MiB]"""
import os, sys
os.environ["MLSD_WSTREAM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import engine
model = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
lat = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
mib = int(sys.argv[4]) if len(sys.argv) > 4 else 512
un = engine.Unet(model, lat, lat, n, stream_weights_mib=mib)
for _ in range(3):
    un.ctx.compute(); un.ctx.sync()
