"""Numerics estimate (numpy, no GPU): the error of the two correction products of a split-fp16 contraction (w_lo x h_hi + w_hi x h_lo)
when their operands are quantised to OCP e4m3 with per-tensor exponents (the shipped q8 images), to e2m3 (the FP6 operand form of
v_mfma_scale_f32_32x32x64_f8f6f4, which runs at twice the FP8 rate on gfx950) with per-tensor exponents, and to e2m3 with one
E8M0 scale per row and 32-column block.  Output quoted in DESIGN.md section 8."""
import numpy as np
rng=np.random.default_rng(0)
def q_e4m3(x):
    # OCP e4m3fn: 4 exp bits (bias 7), 3 mantissa, max 448, subnormal step 2^-9
    s=np.sign(x); a=np.abs(x)
    a=np.minimum(a,448.0)
    e=np.floor(np.log2(np.maximum(a,1e-30)))
    e=np.maximum(e,-6)
    step=2.0**(e-3)
    return s*np.round(a/step)*step
def q_e2m3(x):
    # e2m3: bias 1, max 7.5, subnormal step 0.125 below 1.0
    s=np.sign(x); a=np.minimum(np.abs(x),7.5)
    e=np.floor(np.log2(np.maximum(a,1e-30)))
    e=np.maximum(e,0)
    step=2.0**(e-3)
    return s*np.minimum(np.round(a/step)*step,7.5)
def split(x):
    hi=x.astype(np.float16).astype(np.float64); lo=(x-hi)
    return hi,lo
K=768; R=2048; N=512
W=rng.standard_normal((R,K))/np.sqrt(K)
for name,h in (("uniform h",rng.uniform(-1,1,(N,K))),("saturating h (o*tanh(c), peaky-like)",np.tanh(rng.standard_normal((N,K))*2.5)*rng.uniform(0,1,(N,K))**0.3)):
    wh,wl=split(W); hh,hl=split(h)
    exact=(wl@hh.T+wh@hl.T)            # the two correction products
    main=wh@hh.T
    # e4m3 with per-tensor exponents as shipped: h8 = hi*2^8, l8 = lo*2^19 ; W: 2^ew with max|W|*2^ew <= 448
    ew=int(np.floor(np.log2(448/np.abs(W).max())))
    c8=(q_e4m3(wl*2.0**(ew+11))@q_e4m3(hh*2.0**8).T*2.0**-(ew+11+8) + q_e4m3(wh*2.0**ew)@q_e4m3(hl*2.0**19).T*2.0**-(ew+19))
    # e2m3 with per-tensor exponents: scale so that the maximum maps below 7.5
    e6w=int(np.floor(np.log2(7.5/np.abs(W).max()))); e6h=2   # |h| < 1 -> *4 <= 4 ... use 2^2 (max 4) or 2^3 clipped
    for e6h in (2,3):
        c6=(q_e2m3(wl*2.0**(e6w+11))@q_e2m3(hh*2.0**e6h).T*2.0**-(e6w+11+e6h) + q_e2m3(wh*2.0**e6w)@q_e2m3(hl*2.0**(e6h+11)).T*2.0**-(e6w+e6h+11))
        print(name,"e2m3 (h scale 2^%d): rms err of corrections / rms main = %.3e"%(e6h,np.sqrt(np.mean((c6-exact)**2))/np.sqrt(np.mean(main**2))))
    # e2m3 with per-32-block scales (E8M0 per row and block)
    def qblk(x,target=7.5):
        xb=x.reshape(x.shape[0],-1,32); m=np.abs(xb).max(-1,keepdims=True); sc=2.0**np.floor(np.log2(target/np.maximum(m,1e-30)))
        return (q_e2m3(xb*sc)/sc).reshape(x.shape)
    cb=(qblk(wl)@qblk(hh).T+qblk(wh)@qblk(hl).T)
    print(name,"e4m3 per-tensor (shipped): %.3e"%(np.sqrt(np.mean((c8-exact)**2))/np.sqrt(np.mean(main**2))), " e2m3 per-block scales: %.3e"%(np.sqrt(np.mean((cb-exact)**2))/np.sqrt(np.mean(main**2))), " no correction at all: %.3e"%(np.sqrt(np.mean(exact**2))/np.sqrt(np.mean(main**2))))
