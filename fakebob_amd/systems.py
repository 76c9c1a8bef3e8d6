"""Speaker-recognition systems behind the reference's `model` plugin API (README.md:136):
`score(audios, fs, bits_per_sample, debug, n_jobs)` and `make_decisions(...)`, attributes
`spk_ids` / `threshold`.  Same class names, constructor arguments, return shapes and decision
rules as the reference wrappers (gmm_ubm_OSI.py, gmm_ubm_CSI.py, gmm_ubm_SV.py); the scoring
itself (wav -> MFCC -> VAD -> deltas -> CMVN -> GMM log-likelihoods) runs on the GPU through
libfakebob_hip.so instead of >= 10 Kaldi subprocesses per call.  `n_jobs`, `debug` and `fs` are
accepted and ignored; `bits_per_sample` drives the int16 cast exactly as in the reference.
"""
import os
import warnings

import numpy as np

from .engine import Engine
from .kaldi_io import load_gmm_any


# What the reference's pipeline does to the numbers and the engine reproduces on the device when asked:
#   compress_feats  `copy-feats --compress=true` inside steps/make_mfcc.sh (gmm_ubm_kaldiHelper.py:138-140)
#   text_scores     the 6-significant-digit score text the helpers parse (gmm_ubm_kaldiHelper.py:236-248)
#   mfcc_f32        compute-mfcc-feats in Kaldi's own BaseFloat = float32 arithmetic (k_mfcc_f32)
#                   instead of float64 between Kaldi's float32 storage points
# Every system class carries its own default in `PIPELINE` -- None for the library classes below (full precision: the
# engine's flags are left as the caller set them), REFERENCE_PIPELINE for the subclasses the drop-in modules export
# under the reference's names (fakebob_amd/dropin/).  Nothing process-global is switched by an import.
REFERENCE_PIPELINE = {"text_scores": True, "compress_feats": True, "mfcc_f32": True}
_PIPELINE_ENV = (("text_scores", "FB_TEXT_SCORES"), ("compress_feats", "FB_COMPRESS_FEATS"), ("mfcc_f32", "FB_MFCC_F32"))


def _pipeline_options(text_scores, compress_feats, default=None, mfcc_f32=None):
    """Front-end overrides for the pipeline flags.  Precedence per flag: constructor keyword (True or False), then
    FB_TEXT_SCORES / FB_COMPRESS_FEATS / FB_MFCC_F32 = 0 | 1, then the class default (`default`, a dict or None).  A flag
    that none of the three names is NOT returned: constructing a library system on a shared engine leaves whatever
    the caller set through Engine.set_frontend alone."""
    out = {}
    kws = {"text_scores": text_scores, "compress_feats": compress_feats, "mfcc_f32": mfcc_f32}
    for key, env in _PIPELINE_ENV:
        kw = kws[key]
        if kw is not None:
            out[key] = int(bool(kw))
            continue
        ev = os.environ.get(env)
        if ev in ("0", "1"):
            out[key] = int(ev == "1")
        elif default is not None and key in default:
            out[key] = int(bool(default[key]))
    return out


def _apply_frontend(engine, conf_over, text_scores, compress_feats, mfcc_f32, default):
    """Front-end options of a system under construction -> the engine.  `conf_over`: what pre_model_dir/conf says.
    A float32 MFCC mode that only the class default asked for (no keyword, no FB_MFCC_F32) falls back to the float64
    kernel with a warning when the configuration is outside k_mfcc_f32's shape (raw-energy=false, > 31 mel bins, an odd
    frame length, a padded length other than 512): such a conf worked before the drop-in classes defaulted to float32."""
    from ._native import NativeError
    opts = _pipeline_options(text_scores, compress_feats, default, mfcc_f32)
    over = dict(conf_over, **opts)
    if not over:
        inherited = [k for k, _ in _PIPELINE_ENV if getattr(getattr(engine, "cfg", None), k, 0)]
        if inherited:
            warnings.warn("this system is built on an engine whose pipeline flags %s were switched on earlier (by a drop-in "
                          "system sharing it) and names none of them: it inherits them -- pass text_scores / compress_feats / "
                          "mfcc_f32 = False for the library's full-precision defaults" % inherited)
        return
    only_default = (opts.get("mfcc_f32") == 1 and mfcc_f32 is None and os.environ.get("FB_MFCC_F32") not in ("0", "1"))
    try:
        engine.set_frontend(**over)
    except NativeError:
        if not only_default:
            raise
        over["mfcc_f32"] = 0
        engine.set_frontend(**over)  # (raises again when the float32 mode was not the reason)
        warnings.warn("this front-end configuration is outside the float32 MFCC kernel's shape (it needs padded length 512, "
                      "raw-energy, <= 31 mel bins, <= 32 cepstra, an even frame length): the float64 kernel is used instead")
    if "mfcc_f32" not in opts or "text_scores" not in opts or "compress_feats" not in opts:
        inherited = [k for k, _ in _PIPELINE_ENV if k not in opts and getattr(getattr(engine, "cfg", None), k, 0)]
        if inherited:
            warnings.warn("pipeline flags %s stay on from an earlier system on this engine" % inherited)


def reference_pipeline(cls, module=None):
    """The subclass of a system class that behaves like the reference's pipeline by default (both round trips on):
    what fakebob_amd/dropin/<reference module name>.py exports under the reference's class name.  `module`: the
    exporting module's __name__ -- the subclass then pickles by reference as <module>.<class name>."""
    return type(cls.__name__, (cls,), {"PIPELINE": dict(REFERENCE_PIPELINE), "__doc__": cls.__doc__,
                                       "__module__": module or cls.__module__, "__qualname__": cls.__name__})


def default_device():
    return int(os.environ.get("FAKEBOB_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def _to_audio_list(audios):
    """Input normalisation of the reference wrappers (gmm_ubm_OSI.py:70-81): an (N,)/(N,1)/(1,N)
    ndarray is ONE utterance, an (N,B) ndarray is B columns, anything else is an iterable of 1-D
    arrays of possibly different lengths.  Inputs are never modified (the reference deep-copies)."""
    if isinstance(audios, np.ndarray):
        if audios.ndim == 1 or (audios.ndim == 2 and (audios.shape[0] == 1 or audios.shape[1] == 1)):
            return [np.ascontiguousarray(audios).reshape(-1)]
        if audios.ndim == 2:
            return [np.ascontiguousarray(audios[:, i]) for i in range(audios.shape[1])]
        raise ValueError("audios must be 1-D or 2-D")
    return [np.asarray(a).reshape(-1) for a in audios]


class _GmmSystem(object):
    task = None
    PIPELINE = None  # class default of the two round trips (see REFERENCE_PIPELINE)

    def _setup(self, group_id, models, spk_ids, utt_ids, locations, z_means, z_stds, pre_model_dir, engine,
               text_scores=None, compress_feats=None, mfcc_f32=None):
        self.pre_model_dir = os.path.abspath(pre_model_dir)
        self.group_id = os.path.abspath(group_id)
        self.spk_ids = spk_ids
        self.utt_ids = utt_ids
        self.identity_locations = locations
        self.n_speakers = len(spk_ids)
        self._engine = engine if engine is not None else Engine(default_device())
        conf = os.path.join(self.pre_model_dir, "conf")
        over = {}
        if os.path.isdir(conf):
            from .config import frontend_from_kaldi_conf
            over = frontend_from_kaldi_conf(self.pre_model_dir)
        _apply_frontend(self._engine, over, text_scores, compress_feats, mfcc_f32, self.PIPELINE)
        self._engine.load_gmm(models)
        self._engine.set_system(self.task, z_means, z_stds)

    @property
    def engine(self):
        return self._engine

    def score_utterance(self, audio, fs=16000, bits_per_sample=16, debug=False, n_jobs=5):
        """One utterance -> its score(s): the name BASELINE.json's north_star uses for the scoring surface.  The
        reference has no such symbol (its surface is model.score, README.md:136); this is score() for one audio."""
        return self.score(np.asarray(audio).reshape(-1), fs=fs, bits_per_sample=bits_per_sample, debug=debug,
                          n_jobs=n_jobs)

    def _raw(self, audios, bits_per_sample):
        lst = _to_audio_list(audios)
        raw, _tv = self._engine.score_raw(lst, bits_per_sample=bits_per_sample)
        return raw


class gmm_OSI(_GmmSystem):
    """gmm_ubm_OSI.py:13-112"""
    task = "OSI"

    def __init__(self, group_id, model_list, ubm, pre_model_dir="pre-models", threshold=0.0, engine=None,
                 text_scores=None, compress_feats=None, mfcc_f32=None):
        self.threshold = threshold
        locs = [m[2] for m in model_list]
        self.model_list = [ubm] + locs  # UBM first (gmm_ubm_OSI.py:45)
        models = [load_gmm_any(x) for x in self.model_list]
        self._setup(group_id, models, [m[0] for m in model_list], [m[1] for m in model_list], locs, None, None,
                    pre_model_dir, engine, text_scores, compress_feats, mfcc_f32)

    def score(self, audios, fs=16000, bits_per_sample=16, debug=False, n_jobs=5):
        raw = self._raw(audios, bits_per_sample)
        final = raw[:, 1:] - raw[:, 0:1]                     # :89
        return final if final.shape[0] > 1 else final[0]     # :91

    def make_decisions(self, audios, fs=16000, bits_per_sample=16, n_jobs=5, debug=False):
        score = self.score(audios, fs=fs, bits_per_sample=bits_per_sample, debug=debug, n_jobs=n_jobs)
        if score.ndim == 1:
            score = score[np.newaxis, :]
        max_score = np.max(score, axis=1)
        decisions = list(np.argmax(score, axis=1))
        for i, v in enumerate(max_score):
            if v < self.threshold:                           # strict < (:105)
                decisions[i] = -1
        if score.shape[0] == 1:
            return decisions[0], score.flatten()
        return decisions, score


class gmm_CSI(_GmmSystem):
    """gmm_ubm_CSI.py:13-110 -- no UBM, z-normalised raw log-likelihoods."""
    task = "CSI"

    def __init__(self, group_id, model_list, pre_model_dir="pre-models", engine=None, text_scores=None,
                 compress_feats=None, mfcc_f32=None):
        locs = [m[2] for m in model_list]
        self.model_list = locs
        self.z_norm_means = np.array([m[3] for m in model_list], np.float64)
        self.z_norm_stds = np.array([m[4] for m in model_list], np.float64)
        models = [load_gmm_any(x) for x in locs]
        self._setup(group_id, models, [m[0] for m in model_list], [m[1] for m in model_list], locs,
                    self.z_norm_means, self.z_norm_stds, pre_model_dir, engine, text_scores, compress_feats, mfcc_f32)

    def score(self, audios, fs=16000, bits_per_sample=16, debug=False, n_jobs=5):
        raw = self._raw(audios, bits_per_sample)
        final = (raw - self.z_norm_means) / self.z_norm_stds  # :93
        return final if final.shape[0] > 1 else final[0]

    def make_decisions(self, audios, fs=16000, bits_per_sample=16, n_jobs=5, debug=False):
        score = self.score(audios, fs=fs, bits_per_sample=bits_per_sample, debug=debug, n_jobs=n_jobs)
        if score.ndim == 1:
            score = score[np.newaxis, :]
        decisions = list(np.argmax(score, axis=1))
        if score.shape[0] == 1:
            return decisions[0], score.flatten()
        return decisions, score


class gmm_SV(_GmmSystem):
    """gmm_ubm_SV.py:13-92 -- one enrolled speaker against the UBM."""
    task = "SV"

    def __init__(self, spk_id, model, ubm, pre_model_dir="pre-models", threshold=0.0, engine=None,
                 text_scores=None, compress_feats=None, mfcc_f32=None):
        self.threshold = threshold
        self.utt_id = model[1]
        self.identity_location = model[2]
        self.model_list = [ubm, self.identity_location]
        models = [load_gmm_any(x) for x in self.model_list]
        self._setup(spk_id, models, [model[0]], [model[1]], [model[2]], None, None, pre_model_dir, engine,
                    text_scores, compress_feats, mfcc_f32)
        self.spk_id = self.group_id

    def score(self, audios, fs=16000, bits_per_sample=16, debug=False, n_jobs=5):
        raw = self._raw(audios, bits_per_sample)
        final = raw[:, 1] - raw[:, 0]                        # :77
        return final if final.shape[0] > 1 else final[0]     # (B,) or scalar

    def make_decisions(self, audios, fs=16000, bits_per_sample=16, n_jobs=5, debug=False):
        score = self.score(audios, fs=fs, bits_per_sample=bits_per_sample, debug=debug, n_jobs=n_jobs)
        if isinstance(score, np.ndarray):
            decisions = [1 if s >= self.threshold else -1 for s in score]   # >= (:87)
        else:
            decisions = 1 if score >= self.threshold else -1
        return decisions, score


# ------------------------------------------------------------------ i-vector / PLDA
class _IvSystem(object):
    task = None
    PIPELINE = None

    def _setup(self, group_id, model_list, pre_model_dir, engine, system, text_scores=None, compress_feats=None, mfcc_f32=None):
        from .models import IvectorSystem
        self.pre_model_dir = os.path.abspath(pre_model_dir)
        self.group_id = os.path.abspath(group_id)
        self.n_speakers = len(model_list)
        spk_ids = [m[0] for m in model_list]
        utt_ids = [m[1] for m in model_list]
        locs = [m[2] for m in model_list]
        zm = np.array([m[3] for m in model_list], np.float64)
        zs = np.array([m[4] for m in model_list], np.float64)
        # speakers are re-ordered by sorted spk_id string (ivector_PLDA_OSI.py:56-57,65-82): label
        # indices therefore differ from the GMM wrappers' command-line order
        if len(model_list) > 1:
            order = []
            for sid in sorted(spk_ids):
                order.append(int(np.argwhere(np.array(spk_ids) == sid).flatten()[0]))
            spk_ids = sorted(spk_ids)
            utt_ids = [utt_ids[i] for i in order]
            locs = [locs[i] for i in order]
            zm, zs = zm[order], zs[order]
        self.spk_ids, self.utt_ids, self.identity_locations = spk_ids, utt_ids, locs
        self.z_norm_means, self.z_norm_stds = zm, zs
        self._engine = engine if engine is not None else Engine(default_device())
        from .kaldi_io import load_ivector_pre_models, read_ivector_location
        enrolled = np.stack([read_ivector_location(x) for x in locs])
        if system is None:
            conf = os.path.join(self.pre_model_dir, "conf")
            over = {}
            if os.path.isdir(conf):
                from .config import frontend_from_kaldi_conf
                over = frontend_from_kaldi_conf(self.pre_model_dir)
            _apply_frontend(self._engine, over, text_scores, compress_feats, mfcc_f32, self.PIPELINE)
            d = load_ivector_pre_models(self.pre_model_dir)
            system = IvectorSystem(enrolled=enrolled, z_mean=zm, z_std=zs, **d)
        else:
            _apply_frontend(self._engine, {}, text_scores, compress_feats, mfcc_f32, self.PIPELINE)
            system = system.with_enrolled(enrolled, zm, zs)
        self._engine.load_ivector(system, self.task)

    @property
    def engine(self):
        return self._engine

    def score_utterance(self, audio, fs=16000, bits_per_sample=16, n_jobs=10, debug=False):
        """One utterance -> its score(s): the name BASELINE.json's north_star uses for the scoring surface.  The
        reference has no such symbol (its surface is model.score, README.md:136); this is score() for one audio."""
        return self.score(np.asarray(audio).reshape(-1), fs=fs, bits_per_sample=bits_per_sample, debug=debug,
                          n_jobs=n_jobs)

    def _llr(self, audios, bits_per_sample):
        raw, _tv = self._engine.score_raw(_to_audio_list(audios), bits_per_sample=bits_per_sample)
        return raw


class iv_OSI(_IvSystem):
    """ivector_PLDA_OSI.py:16-143"""
    task = "OSI"

    def __init__(self, group_id, model_list, pre_model_dir="pre-models", threshold=0.0, engine=None, system=None,
                 text_scores=None, compress_feats=None, mfcc_f32=None):
        self.threshold = threshold
        self._setup(group_id, model_list, pre_model_dir, engine, system, text_scores, compress_feats, mfcc_f32)

    def score(self, audio_list, fs=16000, bits_per_sample=16, n_jobs=10, debug=False):
        s = (self._llr(audio_list, bits_per_sample) - self.z_norm_means) / self.z_norm_stds   # :119
        if self.n_speakers == 1:
            return s[:, 0]                     # one enrolled speaker: the helper returns (B,) (:293-296)
        return s if s.shape[0] > 1 else s[0]   # the helper returns (S,) for a single utterance (:282-296)

    def make_decisions(self, audios, fs=16000, bits_per_sample=16, n_jobs=10, debug=False):
        score_array = self.score(audios, fs=fs, bits_per_sample=bits_per_sample, n_jobs=n_jobs, debug=debug)
        if score_array.ndim == 1:
            score_array = score_array[np.newaxis, :]
        max_score = np.max(score_array, axis=1)
        decisions = np.argmax(score_array, axis=1)
        for i, sc in enumerate(max_score):
            if sc < self.threshold:
                decisions[i] = -1
        decisions = list(decisions)
        if len(decisions) == 1:
            return decisions[0], score_array.flatten()
        return decisions, score_array


class iv_CSI(_IvSystem):
    """ivector_PLDA_CSI.py:18-135"""
    task = "CSI"

    def __init__(self, group_id, model_list, pre_model_dir="pre-models", engine=None, system=None, text_scores=None,
                 compress_feats=None, mfcc_f32=None):
        self._setup(group_id, model_list, pre_model_dir, engine, system, text_scores, compress_feats, mfcc_f32)

    def score(self, audio_list, fs=16000, bits_per_sample=16, n_jobs=10, debug=False):
        s = (self._llr(audio_list, bits_per_sample) - self.z_norm_means) / self.z_norm_stds
        if self.n_speakers == 1:
            return s[:, 0]
        return s if s.shape[0] > 1 else s[0]

    def make_decisions(self, audios, fs=16000, bits_per_sample=16, n_jobs=10, debug=False):
        score_array = self.score(audios, fs=fs, bits_per_sample=bits_per_sample, n_jobs=n_jobs, debug=debug)
        if score_array.ndim == 1:
            score_array = score_array[np.newaxis, :]
        decisions = list(np.argmax(score_array, axis=1))
        if len(decisions) == 1:
            return decisions[0], score_array.flatten()
        return decisions, score_array


class iv_SV(_IvSystem):
    """ivector_PLDA_SV.py:20-110"""
    task = "SV"

    def __init__(self, spk_id, model, pre_model_dir="pre-models", threshold=0.0, engine=None, system=None,
                 text_scores=None, compress_feats=None, mfcc_f32=None):
        self.threshold = threshold
        self._setup(spk_id, [model], pre_model_dir, engine, system, text_scores, compress_feats, mfcc_f32)
        self.spk_id = self.group_id
        self.utt_id = model[1]
        self.identity_location = model[2]
        self.z_norm_mean, self.z_norm_std = model[3], model[4]

    def score(self, audio_list, fs=16000, bits_per_sample=16, n_jobs=10, debug=False):
        s = (self._llr(audio_list, bits_per_sample)[:, 0] - self.z_norm_mean) / self.z_norm_std   # :85
        return s if s.size > 1 else s[0]   # (B,) or scalar (:87)

    def make_decisions(self, audio_list, fs=16000, bits_per_sample=16, n_jobs=10, debug=False):
        scores = self.score(audio_list, fs=fs, bits_per_sample=bits_per_sample, n_jobs=n_jobs, debug=debug)
        if isinstance(scores, np.ndarray):
            decisions = [1 if s >= self.threshold else -1 for s in scores]
        else:
            decisions = 1 if scores >= self.threshold else -1
        return decisions, scores

    def make_decisions_value(self, score):
        return -1 if score < self.threshold else 1
