import ctypes, numpy as np, torch, torch.nn.functional as F, sys, time
lib = ctypes.CDLL("./libamxconv.so")
def emul(x, w, b, stride, pad, order, pad_rb=False):
    # x [B,IC,H,W] bf16-exact float; w [OC,IC,KH,KW]; returns [B,OC,OH,OW] float (bf16-rounded)
    B, IC, H, W = x.shape; OC, _, KH, KW = w.shape
    OH = (H + 2*pad - KH)//stride + 1; OW = (W + 2*pad - KW)//stride + 1
    xn = np.ascontiguousarray(x.permute(0,2,3,1).numpy(), dtype=np.float32)
    wn = np.ascontiguousarray(w.permute(2,3,1,0).numpy(), dtype=np.float32)
    bn = np.ascontiguousarray(b.numpy(), dtype=np.float32)
    y = np.zeros((B,OH,OW,OC), dtype=np.uint16)
    fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.conv_emul(fp(xn), fp(wn), fp(bn), fp(y), B,H,W,IC,OC,KH,KW,stride,pad,OH,OW,order, None)
    yt = torch.from_numpy(y.astype(np.int32) << 16).view(torch.float32)
    return yt.permute(0,3,1,2)
if __name__ == "__main__":
    torch.manual_seed(7)
    bf = lambda t: t.bfloat16().float()
    shapes = [("conv_in",3,128,256,3,1,1),("128@256",128,128,256,3,1,1),("down128",128,128,257,3,2,0),("128->256@128",128,256,128,3,1,1),("256@128",256,256,128,3,1,1),
              ("sc128->256",128,256,128,1,1,0),("down256",256,256,129,3,2,0),("256->512@64",256,512,64,3,1,1),("512@64",512,512,64,3,1,1),("sc256->512",256,512,64,1,1,0),
              ("down512",512,512,65,3,2,0),("512@32",512,512,32,3,1,1),("attn1x1",512,512,32,1,1,0),("conv_out",512,32,32,3,1,1)]
    Bs = [int(a) for a in sys.argv[1:]] or [1]
    for B in Bs:
      for (name,IC,OC,H,ks,s,p) in shapes:
        x = bf(torch.randn(B,IC,H,H)); w = bf(torch.randn(OC,IC,ks,ks)*(1.0/(IC*ks*ks))**0.5); b = bf(torch.randn(OC)*0.1)
        y = F.conv2d(x.bfloat16(), w.bfloat16(), b.bfloat16(), stride=s, padding=p).float()
        res = []
        for order in ((2,) if IC == 3 else ((0,1,3) if s==2 else (0,))):
            t0=time.time(); e = emul(x, w, b, s, p, order); dt=time.time()-t0
            res.append((order, int((e != y).sum()), y.numel(), round(dt,1)))
        print(f"B={B} {name}: ", res, flush=True)
