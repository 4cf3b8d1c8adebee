"""ctypes binding of libmrgan_hip.so (include/mrgan_abi.h).

PyTorch is used here only for device memory (the workspace and I/O tensors), the current HIP stream
and, in dist.py, torch.distributed.  There is no CPU or eager-PyTorch fallback: if the HIP library
is missing or fails, every entry point raises.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmrgan_hip.so")

F32, BF16, FP8 = 0, 1, 2
NET_G, NET_D = 0, 1
FLAG_SYNC_STATS, FLAG_FLAT_GRADS, FLAG_GRAPH, FLAG_GRAD_BF16 = 1, 2, 4, 8
D_GEN, D_MAIN, D_ADAM = 0, 1, 2
G_GEN, G_FEAT, G_BWD, G_TAIL, G_ADAM = 0, 1, 2, 3, 4
TUNE_CHAIN, TUNE_KC_CFG, TUNE_KC_PIPE, TUNE_KS_W8, TUNE_KS_GROUP, TUNE_PAIR_GEN, TUNE_HEAD_MFMA = range(7)
REGION_BN_STATS, REGION_FM_MOMENTS, REGION_BN_BWD, REGION_GRAD_D, REGION_GRAD_G, REGION_WORKSPACE = range(6)
REGION_GRAD_D_BF16, REGION_GRAD_G_BF16, REGION_TAIL_D, REGION_TAIL_G = 6, 7, 8, 9

EXPORTS = [
    "mrgan_default_config", "mrgan_workspace_bytes", "mrgan_create", "mrgan_destroy", "mrgan_last_error",
    "mrgan_num_tensors", "mrgan_tensor_shape", "mrgan_set_weights", "mrgan_get_weights", "mrgan_get_slot",
    "mrgan_set_slot", "mrgan_get_iterations", "mrgan_set_iterations", "mrgan_disc_step", "mrgan_gen_step",
    "mrgan_train_pair", "mrgan_sup_step", "mrgan_fp8_calibration", "mrgan_logmel", "mrgan_logmel_frames", "mrgan_region", "mrgan_eval_error", "mrgan_predict_logits", "mrgan_read_metrics",
    "mrgan_pair_hint", "mrgan_set_tuning", "mrgan_debug_noise", "mrgan_debug_tr_probe", "mrgan_debug_gemm", "mrgan_profile_begin", "mrgan_profile_end", "mrgan_debug_ablate", "mrgan_debug_gemm_time", "mrgan_debug_buffer", "mrgan_debug_gemm_fp8",
]
PROF_NAME_LEN = 96


class Config(C.Structure):
    _fields_ = [
        ("d_in", C.c_int32), ("batch", C.c_int32), ("noise_size", C.c_int32),
        ("g_hidden", C.c_int32 * 2), ("d_hidden", C.c_int32 * 5),
        ("num_classes", C.c_int32), ("dtype", C.c_int32),
        ("sigma", C.c_float * 5),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("bn_eps", C.c_float), ("unlabeled_weight", C.c_float),
        ("seed", C.c_uint64),
        ("rank", C.c_int32), ("world", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
    ]


class DiscArgs(C.Structure):
    _fields_ = [
        ("x_lab", C.c_void_p), ("idx_lab", C.c_void_p), ("labels", C.c_void_p),
        ("x_unl", C.c_void_p), ("idx_unl", C.c_void_p), ("z", C.c_void_p),
        ("ld_x_lab", C.c_int64), ("ld_x_unl", C.c_int64),
        ("stream_mode", C.c_int32), ("reserved", C.c_int32),
    ]


class GenArgs(C.Structure):
    _fields_ = [
        ("x_unl", C.c_void_p), ("idx_unl", C.c_void_p), ("z", C.c_void_p),
        ("ld_x_unl", C.c_int64),
        ("stream_mode", C.c_int32), ("reserved", C.c_int32),
    ]


class SupArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("idx", C.c_void_p), ("labels", C.c_void_p),
        ("ld_x", C.c_int64),
        ("stream_mode", C.c_int32), ("rows_valid", C.c_int32),
    ]


_lib = None


def load_library(path=None):
    """Load the HIP library; raises (never falls back) when it is absent.  `path` (first call only) selects another build
    of the same library, e.g. the diagnostic one with in-kernel stamps (make STAMPS=1)."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    if path is not None:
        LIB_PATH = path
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmrgan_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the training path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.mrgan_last_error.restype = C.c_char_p
    for name in EXPORTS:
        if name != "mrgan_last_error":
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


class MrganError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise MrganError("libmrgan_hip error %d: %s" % (rc, load_library().mrgan_last_error().decode()))


def default_config(d_in, batch):
    cfg = Config()
    _check(load_library().mrgan_default_config(C.byref(cfg), int(d_in), int(batch)))
    return cfg


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, dev):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
    return t.to(device=dev, dtype=torch.float32).contiguous()


class Engine(object):
    """One handle = the Keras shared state of one mr_gan() call (weights, Adam slots, iteration counter)
    plus the three compiled functions of mr_gan.py:169-171."""

    def __init__(self, cfg, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("mr_gan_amd needs a HIP device; there is no CPU path")
        self.lib = load_library()
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        nbytes = C.c_size_t()
        _check(self.lib.mrgan_workspace_bytes(C.byref(cfg), C.byref(nbytes)))
        self.workspace = torch.empty(nbytes.value + 512, dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()
        self._ws_off = (-base) % 256
        self.handle = C.c_void_p()
        _check(self.lib.mrgan_create(C.byref(cfg), C.c_void_p(base + self._ws_off), C.c_size_t(nbytes.value), _stream(),
                                     C.byref(self.handle)))
        self._keep = []

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            torch.cuda.synchronize(self.device)
            self.lib.mrgan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights (Keras order) ---------------------------------------------------------------------
    def num_tensors(self, net):
        n = C.c_int()
        _check(self.lib.mrgan_num_tensors(self.handle, net, C.byref(n)))
        return n.value

    def tensor_shape(self, net, idx):
        r, c = C.c_int(), C.c_int()
        _check(self.lib.mrgan_tensor_shape(self.handle, net, idx, C.byref(r), C.byref(c)))
        return r.value, c.value

    def set_weights(self, net, tensors):
        for i, t in enumerate(tensors):
            r, c = self.tensor_shape(net, i)
            t = _f32(t, self.device)
            if t.numel() != r * c:
                raise ValueError("tensor %d of net %d: expected %d x %d, got %s" % (i, net, r, c, tuple(t.shape)))
            _check(self.lib.mrgan_set_weights(self.handle, net, i, _ptr(t), _stream()))
            self._keep.append(t)
        torch.cuda.current_stream().synchronize()
        self._keep = []

    def _fetch(self, net, fn, *extra):
        out = []
        for i in range(self.num_tensors(net)):
            r, c = self.tensor_shape(net, i)
            t = torch.empty(r * c, dtype=torch.float32, device=self.device)
            _check(fn(self.handle, net, i, *extra, _ptr(t), _stream()))
            full = self.full_shape(net, i)
            out.append(t.cpu().numpy().reshape(full))
        return out

    def full_shape(self, net, idx):
        r, c = self.tensor_shape(net, idx)
        is_matrix = (net == NET_D and idx % 2 == 0) or (net == NET_G and idx in (0, 4, 6))
        return (r, c) if is_matrix else (c,)

    def get_weights(self, net):
        return self._fetch(net, self.lib.mrgan_get_weights)

    def get_slot(self, net, which):
        return self._fetch(net, self.lib.mrgan_get_slot, which)

    def set_slot(self, net, which, tensors):
        for i, t in enumerate(tensors):
            t = _f32(t, self.device)
            _check(self.lib.mrgan_set_slot(self.handle, net, i, which, _ptr(t), _stream()))
            self._keep.append(t)
        torch.cuda.current_stream().synchronize()
        self._keep = []

    def get_iterations(self):
        it = C.c_uint32()
        _check(self.lib.mrgan_get_iterations(self.handle, _stream(), C.byref(it)))
        return it.value

    def set_iterations(self, iterations, batch_counter=0):
        _check(self.lib.mrgan_set_iterations(self.handle, C.c_uint32(iterations), C.c_uint32(batch_counter), _stream()))

    # ---- the compiled functions -------------------------------------------------------------------------
    @staticmethod
    def disc_args(x_lab, labels, x_unl, z=None, idx_lab=None, idx_unl=None, stream_mode=0):
        a = DiscArgs()
        a.x_lab, a.idx_lab, a.labels = _ptr(x_lab), _ptr(idx_lab), _ptr(labels)
        a.x_unl, a.idx_unl, a.z = _ptr(x_unl), _ptr(idx_unl), _ptr(z)
        a.ld_x_lab, a.ld_x_unl = x_lab.stride(0), x_unl.stride(0)
        a.stream_mode = stream_mode
        a._refs = (x_lab, labels, x_unl, z, idx_lab, idx_unl)    # the struct holds raw pointers: keep the tensors alive
        return a

    @staticmethod
    def gen_args(x_unl, z=None, idx_unl=None, stream_mode=0):
        a = GenArgs()
        a.x_unl, a.idx_unl, a.z = _ptr(x_unl), _ptr(idx_unl), _ptr(z)
        a.ld_x_unl = x_unl.stride(0)
        a.stream_mode = stream_mode
        a._refs = (x_unl, z, idx_unl)
        return a

    def disc_step(self, args, first=0, last=-1, want_outputs=True):
        """train_batch_disc -> (loss_lab, loss_unl, train_err)"""
        out = (C.c_float * 3)()
        _check(self.lib.mrgan_disc_step(self.handle, C.byref(args), first, last, out if want_outputs else None, _stream()))
        return tuple(out) if want_outputs else None

    def gen_step(self, args, first=0, last=-1, want_outputs=True):
        """train_batch_gen -> loss_gen"""
        out = (C.c_float * 1)()
        _check(self.lib.mrgan_gen_step(self.handle, C.byref(args), first, last, out if want_outputs else None, _stream()))
        return out[0] if want_outputs else None

    @staticmethod
    def sup_args(x, labels, idx=None, stream_mode=0, rows_valid=0):
        a = SupArgs()
        a.x, a.idx, a.labels = _ptr(x), _ptr(idx), _ptr(labels)
        a.ld_x = x.stride(0)
        a.stream_mode = stream_mode
        a.rows_valid = rows_valid
        a._refs = (x, labels, idx)
        return a

    def sup_step(self, args, want_outputs=True):
        """one supervised train_on_batch of the NN baseline (mr_nn.py:117) -> (mse loss, training error of the batch)"""
        out = (C.c_float * 2)()
        _check(self.lib.mrgan_sup_step(self.handle, C.byref(args), out if want_outputs else None, _stream()))
        return tuple(out) if want_outputs else None

    FP8_DRY_PASSES = 5
    FP8_CAL_QUERY, FP8_CAL_BEGIN, FP8_CAL_END_PASS, FP8_CAL_DONE = 0, 1, 2, 3

    def fp8_calibration(self, kind, action):
        """phase-wise hosts (mr_gan_amd/dist.py): settle the fp8 scales of sub-step kind 0 (D) / 1 (G); see mrgan_abi.h"""
        rc = self.lib.mrgan_fp8_calibration(self.handle, int(kind), int(action), _stream())
        if rc < 0:
            _check(rc)
        return rc

    def train_pair(self, dargs, gargs):
        _check(self.lib.mrgan_train_pair(self.handle, C.byref(dargs), C.byref(gargs), _stream()))

    def set_tuning(self, knob, value):
        """launch-structure knobs (TUNE_*): results stay within rounding"""
        _check(self.lib.mrgan_set_tuning(self.handle, int(knob), int(value)))

    def pair_hint(self, on=True):
        """the next disc_step is followed by a gen_step with device-drawn z: share the generator pass (no-op with synced statistics)"""
        _check(self.lib.mrgan_pair_hint(self.handle, 1 if on else 0))

    def eval_error(self, x, labels, idx=None):
        """test_batch -> err"""
        err = C.c_float()
        _check(self.lib.mrgan_eval_error(self.handle, _ptr(x), _ptr(idx), C.c_int64(x.stride(0)), _ptr(labels),
                                         C.c_int64(labels.numel() if idx is None else idx.numel()), C.byref(err), _stream()))
        return err.value

    def predict_logits(self, x, idx=None):
        n = x.shape[0] if idx is None else idx.numel()
        out = torch.empty((n, self.cfg.num_classes), dtype=torch.float32, device=self.device)
        _check(self.lib.mrgan_predict_logits(self.handle, _ptr(x), _ptr(idx), C.c_int64(x.stride(0)), C.c_int64(n), _ptr(out),
                                             _stream()))
        return out

    def read_metrics(self, reset=True):
        out = (C.c_float * 8)()
        _check(self.lib.mrgan_read_metrics(self.handle, out, 1 if reset else 0, _stream()))
        return list(out)

    def profile_begin(self):
        _check(self.lib.mrgan_profile_begin(self.handle))

    def profile_end(self, max_kernels=64):
        """-> {kernel instantiation name: (total ms, launches, algorithmic flops, algorithmic bytes)} since profile_begin"""
        names = C.create_string_buffer(max_kernels * PROF_NAME_LEN)
        ms, cnt = (C.c_float * max_kernels)(), (C.c_int32 * max_kernels)()
        fl, by, n = (C.c_double * max_kernels)(), (C.c_double * max_kernels)(), C.c_int()
        _check(self.lib.mrgan_profile_end(self.handle, _stream(), max_kernels, names, ms, cnt, fl, by, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = names.raw[i * PROF_NAME_LEN:(i + 1) * PROF_NAME_LEN].split(b"\0")[0].decode()
            if name != "(start)":
                out[name] = (ms[i], cnt[i], fl[i], by[i])
        return out

    def region(self, which):
        """torch view of a workspace region (aliases library memory: used for all-reduce): fp32, bfloat16 for the
        REGION_GRAD_*_BF16 regions of a FLAG_GRAD_BF16 handle."""
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.lib.mrgan_region(self.handle, which, C.byref(p), C.byref(n)))
        off = p.value - self.workspace.data_ptr()
        return self.workspace[off:off + n.value].view(torch.bfloat16 if which in (REGION_GRAD_D_BF16, REGION_GRAD_G_BF16) else torch.float32)

    def debug_ablate(self, bits):
        _check(self.lib.mrgan_debug_ablate(self.handle, int(bits)))

    def debug_buffer(self, kind, l, nseg=3):
        """activation buffer (0 xin[l], 1 dpre[l], 2 features) as a float32 tensor [nseg, S, ld] (a copy)"""
        p, rows, ld, es = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        _check(self.lib.mrgan_debug_buffer(self.handle, kind, l, C.byref(p), C.byref(rows), C.byref(ld), C.byref(es)))
        off = p.value - self.workspace.data_ptr()
        n = nseg * rows.value * ld.value * es.value
        raw = self.workspace[off:off + n]
        t = raw.view(torch.bfloat16 if es.value == 2 else torch.float32)
        return t.reshape(nseg, rows.value, ld.value).float().clone()

    def debug_noise(self, site, seg, step, rows, cols, row0=0):
        out = torch.empty((rows, cols), dtype=torch.float32, device=self.device)
        _check(self.lib.mrgan_debug_noise(self.handle, site, seg, step, row0, rows, cols, _ptr(out), _stream()))
        return out


def debug_gemm(dtype, op, a, b, bias=None, act=0, splits=1):
    """Raw kernel-level product for parity tests. op 0: act(a b + bias); 1: a b^T; 2: a^T b."""
    lib = load_library()
    m = a.shape[0]
    if op == 0:
        k, n = b.shape
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    elif op == 1:
        k, n = b.shape
        out = torch.empty((m, k), dtype=torch.float32, device=a.device)
    else:
        k, n = a.shape[1], b.shape[1]
        out = torch.empty((k, n), dtype=torch.float32, device=a.device)
    _check(lib.mrgan_debug_gemm(dtype, op, m, n, k, _ptr(a), _ptr(b), _ptr(bias), act, splits, _ptr(out), _stream()))
    return out


def debug_tr_probe(device="cuda:0"):
    out = torch.zeros(1024, dtype=torch.int16, device=device)
    _check(load_library().mrgan_debug_tr_probe(_ptr(out), _stream()))
    return out.cpu().numpy().astype(np.uint16).reshape(2, 64, 8)


def debug_gemm_time(op, m, n, k, nbatch=1, splits=1, reps=50, ablate=0, kc_cfg=-1):
    us = C.c_float()
    _check(load_library().mrgan_debug_gemm_time(op, m, n, k, nbatch, splits, reps, ablate, kc_cfg, C.byref(us)))
    return us.value


def debug_gemm_fp8(a, b, bias=None, act=0, scale_a=1.0, scale_b=1.0, reps=0, kc_cfg=-1):
    """e4m3 forward product act((q(a sa) q(b sb)) / (sa sb) + bias) -> (fp32 [m, n] result, average us per launch if reps > 0)"""
    m, k = a.shape
    n = b.shape[1]
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    us = C.c_float()
    _check(load_library().mrgan_debug_gemm_fp8(m, n, k, _ptr(a), _ptr(b), _ptr(bias), act, C.c_float(scale_a), C.c_float(scale_b), _ptr(out),
                                               reps, C.byref(us), int(kc_cfg), _stream()))
    return out, us.value
