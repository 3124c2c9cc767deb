"""Exhaustive check behind smr_convert_420.h: for the two range divisors of planar_yuv_to_rgba.wgsl:46-48 and EVERY f32 a in [-0.0628, 0.9374]
(|a| >= 2^-40: what c - 16/255 can be for c in [0, 1]), RN(a * y_hi + RN(a * y_lo)) with y_hi = RN(1 / d), y_lo = RN(1 / d - y_hi) equals the IEEE quotient a / d.
About 25 s of numpy; prints the mismatch count (0) per divisor.  `python tools/check_div_by_constant.py`"""
import numpy as np, sys, time
kc=np.float32(0.87843137254); ky=np.float32(0.85882352941)
def check(d, lo, hi):
    d64=np.float64(d)
    yh=np.float32(1.0)/d
    yl=np.float32(1.0/d64-np.float64(yh))
    bad=0; tot=0
    # all f32 bit patterns between lo and hi (positive range) and the negative range
    ranges=[]
    if hi>0: ranges.append((np.float32(2.0**-40).view(np.uint32), np.float32(hi).view(np.uint32), 0))
    if lo<0: ranges.append((np.float32(2.0**-40).view(np.uint32), np.float32(-lo).view(np.uint32), 0x80000000))
    for b0,b1,sign in ranges:
        b=int(b0)
        while b<=int(b1):
            n=min(1<<24, int(b1)-b+1)
            bits=(np.arange(b,b+n,dtype=np.uint64)|sign).astype(np.uint32)
            a=bits.view(np.float32)
            want=a/d
            p=(a*yl)                      # fl32(a*yl)
            s=a.astype(np.float64)*np.float64(yh)+p.astype(np.float64)
            got=s.astype(np.float32)
            nb=int((got.view(np.uint32)!=want.view(np.uint32)).sum())
            if nb:
                i=np.nonzero(got.view(np.uint32)!=want.view(np.uint32))[0][:3]
                print("mismatch", d, a[i], got[i], want[i]); bad+=nb
            tot+=n; b+=n
    print("divisor",d,"checked",tot,"mismatches",bad, "yh",yh,"yl",yl); sys.stdout.flush()
    return bad
t=time.time()
b1=check(kc, -0.0628, 0.9374)
b2=check(ky, -0.0628, 0.9374)
print("done",time.time()-t,"s")
