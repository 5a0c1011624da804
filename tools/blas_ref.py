import torch, time
torch.backends.cuda.matmul.allow_fp16_reduced_precision_reduction = False
for (M,N,K) in [(8192,8192,8192),(4096,4096,16384),(8192,10240,1280),(8192,1280,5120),(8192,3840,1280),(8192,1280,1280)]:
    a=torch.randn(M,K,device='cuda',dtype=torch.float16); b=torch.randn(N,K,device='cuda',dtype=torch.float16)
    for _ in range(3): c=a@b.t()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    best=1e9
    for r in range(3):
        e0.record()
        for _ in range(10): c=a@b.t()
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/10)
    print(f"hipBLASLt (torch.matmul f16) {M}x{N}x{K}: {best*1e3:8.1f} us {2*M*N*K/best/1e9:8.1f} TF/s", flush=True)
