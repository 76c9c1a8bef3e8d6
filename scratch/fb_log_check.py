# numpy model of fb_log_f64 (fakebob_amd/csrc/fb_device.h) against numpy's log: max 2 ulp, 0.8 % of arguments differ
import numpy as np
def fb_log(x):
    m, e = np.frexp(x)
    lo = m < 0.70710678118654752
    m = np.where(lo, m * 2.0, m); e = np.where(lo, e - 1, e).astype(np.float64)
    s = (m - 1.0) / (m + 1.0); z = s * s
    p = np.full_like(z, 1 / 21.)
    for k in (19, 17, 15, 13, 11, 9, 7, 5, 3):
        p = p * z + 1.0 / k
    lm = 2 * s + 2 * s * z * p
    return e * float.fromhex('0x1.62e42fee00000p-1') + (e * float.fromhex('0x1.a39ef35793c76p-33') + lm)
rng = np.random.default_rng(0)
x = np.exp(rng.uniform(-16, 45, size=3000000))
d = np.abs(fb_log(x) - np.log(x)) / np.spacing(np.abs(np.log(x)))
print("max ulp", d.max(), "fraction differing", (d > 0).mean())
