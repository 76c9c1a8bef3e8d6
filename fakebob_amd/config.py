"""Kaldi option files -> engine front-end options.

The reference hands `pre-models/conf/mfcc.conf`, `conf/vad.conf` and `pre-models/delta_opts` to
Kaldi programs (gmm_ubm_kaldiHelper.py:133,153,191-193); the engine reads the same files so that
nothing about the recipe is hard-coded.  [EXT] option names are Kaldi's.
"""
import os
import re
import warnings


def _parse_opts(text):
    out = {}
    for line in text.splitlines():
        line = line.split("#", 1)[0]
        for m in re.finditer(r"--([A-Za-z0-9_-]+)=(\S+)", line):
            out[m.group(1)] = m.group(2)
    return out


def _b(v):
    return 1 if str(v).lower() in ("true", "t", "1", "yes") else 0


def frontend_overrides(mfcc_conf="", vad_conf="", delta_opts=""):
    """Option text -> dict of fb_frontend_cfg field overrides."""
    o = {}
    m = _parse_opts(mfcc_conf)
    fs = float(m.get("sample-frequency", 16000))
    if "sample-frequency" in m:
        o["sample_freq"] = fs
    if "frame-length" in m:
        o["frame_length"] = int(fs * 0.001 * float(m["frame-length"]))
    if "frame-shift" in m:
        o["frame_shift"] = int(fs * 0.001 * float(m["frame-shift"]))
    if "frame_length" in o or "round-to-power-of-two" in m:
        L = o.get("frame_length", 400)
        if _b(m.get("round-to-power-of-two", "true")):
            p = 1
            while p < L:
                p *= 2
            o["padded_length"] = p
        else:
            raise ValueError("round-to-power-of-two=false is unsupported (FFT size must be a power of two)")
    for k, f, conv in [("low-freq", "low_freq", float), ("high-freq", "high_freq", float),
                       ("num-mel-bins", "num_mel_bins", int), ("num-ceps", "num_ceps", int),
                       ("snip-edges", "snip_edges", _b), ("preemphasis-coefficient", "preemph", float),
                       ("cepstral-lifter", "cepstral_lifter", float), ("remove-dc-offset", "remove_dc", _b),
                       ("use-energy", "use_energy", _b), ("raw-energy", "raw_energy", _b),
                       ("energy-floor", "energy_floor", float)]:
        if k in m:
            o[f] = conv(m[k])
    if "window-type" in m and m["window-type"] != "povey":
        raise ValueError("window-type=%s unsupported (povey only)" % m["window-type"])
    # Kaldi's default is --dither=1.0 and the stock voxceleb mfcc.conf does not override it: a real Kaldi run adds
    # random noise of one LSB to every sample before the MFCC, which no re-implementation can reproduce
    dither = float(m.get("dither", 1.0))
    if dither != 0.0:
        warnings.warn("Kaldi would run with dither=%g (%s): its features are random at the 1-LSB level; the engine "
                      "always uses dither=0, so scores differ from a Kaldi run by that noise"
                      % (dither, "set in mfcc.conf" if "dither" in m else "Kaldi's default, mfcc.conf does not set it"))
    v = _parse_opts(vad_conf)
    for k, f, conv in [("vad-energy-threshold", "vad_energy_threshold", float),
                       ("vad-energy-mean-scale", "vad_energy_mean_scale", float),
                       ("vad-proportion-threshold", "vad_proportion_threshold", float),
                       ("vad-frames-context", "vad_frames_context", int)]:
        if k in v:
            o[f] = conv(v[k])
    d = _parse_opts(delta_opts)
    if "delta-window" in d:
        o["delta_window"] = int(d["delta-window"])
    if "delta-order" in d:
        o["delta_order"] = int(d["delta-order"])
    return o


def frontend_from_kaldi_conf(pre_model_dir):
    def rd(p):
        p = os.path.join(pre_model_dir, p)
        if os.path.isfile(p):
            with open(p) as r:
                return r.read()
        return ""
    return frontend_overrides(rd("conf/mfcc.conf"), rd("conf/vad.conf"), rd("delta_opts"))
