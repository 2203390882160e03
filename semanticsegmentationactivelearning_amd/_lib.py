"""ctypes binding of libssal_hip.so (C ABI declared in include/ssal_enet.h).

The HIP library is the product path.  There is NO CPU fallback: if the shared object is missing
or a call fails, the error is raised loudly.
"""
import ctypes
import os
import sys
import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSAL_LIB_PATH: measurement builds only (tools/phase_trace.py loads a -DSSAL_PHASE_TRACE build)
LIB_PATH = os.environ.get("SSAL_LIB_PATH") or os.path.join(_HERE, "libssal_hip.so")

# ---- process environment the HIP runtime reads ONCE, when it comes up ------------------------------------------------
# GPU_MAX_HW_QUEUES: streams -> hardware queues, runtime default 4 per process.  A ranking job under torch.distributed.run
# drives the caller's stream + 2 image-group side streams + a prefetch copy stream + RCCL's stream = 5; with 2 queues (or an
# EMPTY variable) two chains share a queue and the pass loses 11 % (profiles/r04_ab_hw_queues.txt; 4 = 8 = 16 at N = 1).
# HSA_ENABLE_IPC_MODE_LEGACY=0: the pool's host driver only supports dmabuf IPC (RCCL across the ranks of one node).
# Both are defaulted here, at import, while the runtime is still down; once a HIP context exists they can no longer take
# effect and `warn_if_few_hw_queues` (called by rank_confidence under a process group) says so instead.
MIN_HW_QUEUES = 5
ENV_INJECTED = {}


def _hip_is_up():
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


def _default_runtime_env():
    if _hip_is_up():
        return
    if not os.environ.get("GPU_MAX_HW_QUEUES"):  # unset OR empty (an empty value was measured to behave like 2)
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
        ENV_INJECTED["GPU_MAX_HW_QUEUES"] = "8"
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        ENV_INJECTED["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"


_default_runtime_env()
_warned_hw_queues = False


def hw_queues_ok():
    """True when GPU_MAX_HW_QUEUES is set to >= MIN_HW_QUEUES in this process' environment"""
    v = os.environ.get("GPU_MAX_HW_QUEUES", "")
    try:
        return int(v) >= MIN_HW_QUEUES
    except ValueError:
        return False


def warn_if_few_hw_queues():
    """once per process: the N > 1 ranking loop needs 5 concurrent streams (see above)"""
    global _warned_hw_queues
    if _warned_hw_queues or hw_queues_ok():
        return False
    _warned_hw_queues = True
    warnings.warn("GPU_MAX_HW_QUEUES=%r: a sharded ranking pass drives 5 HIP streams (caller + 2 image-group chains + "
                  "prefetch copy + RCCL); with fewer hardware queues two of them serialise (-11 %% measured).  Export "
                  "GPU_MAX_HW_QUEUES=8, or import semanticsegmentationactivelearning_amd before the first HIP call."
                  % os.environ.get("GPU_MAX_HW_QUEUES"), RuntimeWarning, stacklevel=3)
    return True


SSAL_OK, SSAL_EINVAL, SSAL_EHIP, SSAL_ENOTIMPL, SSAL_ESTATE, SSAL_ENOMEM = range(6)

MEASURES = {"entropy": 0, "margin": 1, "confidence": 2}
# include/ssal_enet.h SSAL_ARITH_*: "f32" = exact fp32, bit-identical to the oracle (default everywhere);
# "bf16x3" = opt-in split-operand bf16 MFMAs in the 128-channel bottlenecks (within north_star's 1e-4, not bit-identical)
ARITHMETICS = {"f32": 0, "bf16x3": 1}


def arithmetic_code(arithmetic):
    if arithmetic not in ARITHMETICS:
        raise ValueError("arithmetic must be one of %s (got %r)" % (sorted(ARITHMETICS), arithmetic))
    return ARITHMETICS[arithmetic]

_c = ctypes
_vp, _i, _i64, _f = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

# symbol -> (restype, argtypes); one row per declaration in include/ssal_enet.h and include/ssal_icnet.h
PROTOTYPES = {
    "ssal_version": (_c.c_char_p, []),
    "ssal_last_error": (_c.c_char_p, []),
    "ssal_enet_create": (_i, [_i, _i, _c.POINTER(_vp)]),
    "ssal_enet_destroy": (_i, [_vp]),
    "ssal_enet_num_tensors": (_i, [_vp]),
    "ssal_enet_tensor_info": (_i, [_vp, _i, _c.POINTER(_c.c_char_p), _c.POINTER(_i), _c.POINTER(_i64)]),
    "ssal_enet_set_tensor": (_i, [_vp, _c.c_char_p, _vp, _i64]),
    "ssal_enet_commit": (_i, [_vp, _vp]),
    "ssal_enet_workspace_bytes": (_i64, [_vp, _i, _i, _i]),
    "ssal_enet_forward_nhwc": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "ssal_enet_score_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ssal_enet_forward_nhwc_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "ssal_enet_score_nhwc_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ssal_enet_forward_nhwc_arith": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "ssal_enet_score_nhwc_arith": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ssal_enet_run_layer_arith": (_i, [_vp, _c.c_char_p, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ssal_enet_endpoint_offsets": (_i, [_vp, _i, _i, _i, _c.POINTER(_i64)]),
    "ssal_enet_export_argmax": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "ssal_enet_run_layer": (_i, [_vp, _c.c_char_p, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ssal_enet_layer_workspace_bytes": (_i64, [_vp, _c.c_char_p, _i, _i, _i]),
    "ssal_score_workspace_bytes": (_i64, [_i, _i, _i]),
    "ssal_score_logits_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ssal_xent_workspace_bytes": (_i64, [_i, _i]),
    "ssal_masked_softmax_cross_entropy": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _i64, _vp]),
    "ssal_max_pool_with_argmax_2x2": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "ssal_unpool_2d": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ssal_prelu": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "ssal_spatial_dropout": (_i, [_vp, _i, _i64, _i, _f, _c.c_uint64, _vp, _vp]),
    "ssal_batch_norm_inference": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssal_conv2d_same": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ssal_conv2d_transpose_3x3_s2": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "ssal_resize_bilinear": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ssal_synth_frames_nhwc": (_i, [_c.c_uint64, _i64, _i, _i, _i, _i, _vp, _vp]),
    "ssal_synth_frames_nhwc_u8": (_i, [_c.c_uint64, _i64, _i, _i, _i, _i, _vp, _vp]),
    "ssal_set_kernel_family": (_i, [_i]),
    "ssal_debug_probe": (_i, [_vp, _vp]),
    "ssal_debug_set_trace": (_i, [_vp, _i64]),
    "ssal_debug_set_knob": (_i, [ctypes.c_char_p, _i]),
    "ssal_debug_get_knobs": (_i, [_c.c_char_p, _i64]),
    # ---- include/ssal_icnet.h ----
    "ssal_icnet_create": (_i, [_i, _i, _c.POINTER(_vp)]),
    "ssal_icnet_destroy": (_i, [_vp]),
    "ssal_icnet_num_tensors": (_i, [_vp]),
    "ssal_icnet_tensor_info": (_i, [_vp, _i, _c.POINTER(_c.c_char_p), _c.POINTER(_i), _c.POINTER(_i64)]),
    "ssal_icnet_set_tensor": (_i, [_vp, _c.c_char_p, _vp, _i64]),
    "ssal_icnet_commit": (_i, [_vp, _vp]),
    "ssal_icnet_workspace_bytes": (_i64, [_vp, _i, _i, _i]),
    "ssal_icnet_forward_nhwc": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "ssal_icnet_forward_nhwc_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "ssal_icnet_score_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ssal_icnet_score_nhwc_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ssal_icnet_num_endpoints": (_i, [_vp]),
    "ssal_icnet_endpoint_name": (_i, [_vp, _i, _c.POINTER(_c.c_char_p)]),
    "ssal_icnet_endpoint_info": (_i, [_vp, _c.c_char_p, _i, _i, _i, _c.POINTER(_i64), _c.POINTER(_i64)]),
    "ssal_icnet_endpoint_valid_after_score": (_i, [_vp, _c.c_char_p, _i, _i]),
    "ssal_conv_bn_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "ssal_conv_bn_act": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                              _vp, _vp, _i64, _vp]),
    "ssal_max_pool_3x3_s2": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "ssal_pyramid_pooling": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "ssal_upscore_workspace_bytes": (_i64, [_i, _i, _i]),
    "ssal_upscore_logits_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    # ---- measurement aids (include/ssal_enet.h) ----
    "ssal_profile_enable": (_i, [_i]),
    "ssal_profile_collect": (_i, [_c.c_char_p, _i64]),
}

# extra entry points of the measurement libraries only (csrc/ssal_measure_api.h; tools/mem_probe.py, tools/mfma_peak.py)
MEASURE_PROTOTYPES = {
    "ssal_debug_mfma_peak": (_i, [_i, _i, _i, _vp, _vp]),
    "ssal_debug_copy_probe": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
}

_LIB = None


def lib():
    """Load libssal_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libssal_hip.so is missing (%s). Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'`. "
                "The MI355X HIP path has no CPU fallback." % LIB_PATH)
        # torch ships its own libamdhip64.so.7 / libhsa-runtime64: it must be mapped BEFORE our
        # library so that both share ONE HIP runtime (loading ours first would bind torch to the
        # system runtime + its bundled HSA and no device would be found).
        _torch()
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in MEASURE_PROTOTYPES.items():
            fn = getattr(handle, name, None)  # present in -DSSAL_MEASURE builds only
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _LIB = handle
    return _LIB


def check(status):
    """Map a C-ABI status to the exception the reference would raise."""
    if status == SSAL_OK:
        return
    msg = lib().ssal_last_error().decode("utf-8", "replace")
    if status == SSAL_EINVAL:
        raise ValueError(msg)
    if status == SSAL_ENOTIMPL:
        raise NotImplementedError(msg)
    if status == SSAL_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError("libssal_hip: status %d: %s" % (status, msg))


def _torch():
    import torch
    return torch


def usable_cores(cap=None):
    """CPU threads this process may really use: min(affinity, cgroup quota[, cap]).  A GPU box
    exposes every host core in the affinity mask but grants a 1-GPU job only a ~16-CPU share;
    sizing thread pools to the mask would oversubscribe it badly."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    env = os.environ.get("SSAL_CPU_THREADS")
    if env:
        n = min(n, max(1, int(env)))
    if cap:
        n = min(n, cap)
    return max(1, n)


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X/HIP device visible: the scoring path runs on the GPU only "
                           "(there is no CPU fallback)")
    return torch


def dev_ptr(t, dtype=None, name="tensor"):
    """Device pointer of a contiguous CUDA(HIP) torch tensor, with dtype / residency checks."""
    torch = _torch()
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError("%s must be a torch tensor resident on the GPU" % name)
    if dtype is not None and t.dtype != dtype:
        raise ValueError("%s must have dtype %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    torch = _torch()
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceState:
    """Device-side state of one Python model object (mixin of models.ENet / models.ICNet).

    * a C handle owns its weights on ONE device (``include/ssal_enet.h``): the model keeps one handle per device ordinal
      and pushes the weights to a device the first time it is used there (and again after any variable changed);
    * concurrent calls on one handle need their own workspace: workspaces are keyed by (device, current torch stream),
      so two host threads scoring on two streams never share one (the header's re-entrancy contract).
    "The most recent call" (endpoint views, ``pooling_argmax``) means the CALLING THREAD's most recent call: ``_ws`` /
    ``_last_dims`` / ``_last_call`` are one record ``(workspace, dims, device, kind)`` kept per thread, so two threads that
    score on one model never pair one's dims with the other's workspace.

    The model provides ``_create_handle(L) -> c_void_p``, ``_push_tensors(L, handle)`` (set_tensor + commit),
    ``_destroy_handle(L, handle)`` and ``variables``."""

    MAX_WORKSPACES = 4  # live (device, stream) workspaces per model; the least recently used one is released beyond that

    def _init_device_state(self):
        import threading
        self._handles = {}        # device ordinal -> [handle, pushed_versions]
        self._workspaces = {}     # (device ordinal, stream pointer) -> uint8 tensor, in least-recently-used order
        self._tls = threading.local()  # .last = (workspace, dims, device ordinal, "forward" | "score" | "layer")
        self._state_lock = threading.Lock()

    # ---- the calling thread's most recent call ----
    def _note_call(self, ws, dims, kind):
        dev = ws.device.index if ws is not None else None
        self._tls.last = (ws, dims, dev, kind)

    @property
    def _last(self):
        return getattr(self._tls, "last", (None, None, None, None))

    @property
    def _ws(self):
        return self._last[0]

    @property
    def _last_dims(self):
        return self._last[1]

    @property
    def _last_call(self):
        return self._last[3]

    @property
    def _handle(self):
        """the handle of the device the calling thread's most recent call ran on (else of the current device); None when
        there is none -- never another device's handle"""
        torch = _torch()
        dev = self._last[2]
        if dev is None and torch.cuda.is_available():
            dev = torch.cuda.current_device()
        ent = self._handles.get(dev)
        return ent[0] if ent else None

    def _sync_handle(self):
        torch = _torch()
        L = lib()
        dev = torch.cuda.current_device()
        with self._state_lock:  # two threads must not create / commit one handle at the same time
            ent = self._handles.get(dev)
            if ent is None:
                ent = self._handles[dev] = [self._create_handle(L), None]
            versions = tuple(v.version for v in self.variables)
            if versions != ent[1]:
                # commit rewrites the weight arena: no call may be in flight on this handle while it runs
                torch.cuda.synchronize(dev)
                self._push_tensors(L, ent[0])
                ent[1] = versions
            return ent[0]

    def _workspace(self, nbytes, device):
        torch = require_gpu()
        key = (device.index if device.index is not None else torch.cuda.current_device(),
               torch.cuda.current_stream(device).cuda_stream)
        with self._state_lock:
            ws = self._workspaces.pop(key, None)
            if ws is None or ws.numel() < nbytes:
                del ws  # release before growing
                # a workspace is 1.8 GB at batch 8 x 1024 x 2048: a caller that cycles through many streams must not pin
                # one per stream for ever.  Dropping the tensor is safe while its stream still runs: the caching
                # allocator hands a block back only to allocations made on the stream it was allocated on
                while len(self._workspaces) >= self.MAX_WORKSPACES:
                    old_key = next(iter(self._workspaces))
                    old = self._workspaces.pop(old_key)
                    del old  # (a thread whose last call used it keeps its own reference until its next call)
                ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._workspaces[key] = ws  # most recently used last
            return ws

    def _release_device_state(self):
        try:
            L = lib()
            for ent in self._handles.values():
                self._destroy_handle(L, ent[0])
            self._handles = {}
        except Exception:
            pass


def as_device_f32(x, device=None):
    """numpy / torch input -> contiguous float32 torch tensor on the current GPU."""
    torch = require_gpu()
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    if not isinstance(x, torch.Tensor):
        raise ValueError("expected a numpy array or torch tensor")
    if x.dtype != torch.float32:
        x = x.float()
    if not x.is_cuda:
        x = x.cuda(device) if device is not None else x.cuda()
    return x.contiguous()


def as_device_image(x, device=None):
    """model input: uint8 stays uint8 (the decoded frame; the Initial block converts it on the fly exactly like
    tf.image.convert_image_dtype), everything else becomes float32; contiguous, on the current GPU."""
    torch = require_gpu()
    is_u8 = (isinstance(x, np.ndarray) and x.dtype == np.uint8) or (isinstance(x, torch.Tensor) and x.dtype == torch.uint8)
    if not is_u8:
        return as_device_f32(x, device)
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not x.is_cuda:
        x = x.cuda(device) if device is not None else x.cuda()
    return x.contiguous()


def profile_enable(on=True):
    check(lib().ssal_profile_enable(1 if on else 0))


def profile_collect():
    """-> {kernel: {"launches", "ms", "flops", "bytes"}} for every launch since the last collect"""
    import json
    buf = ctypes.create_string_buffer(1 << 16)
    check(lib().ssal_profile_collect(buf, len(buf)))
    return json.loads(buf.value.decode())


def set_kernel_family(use_mfma=True):
    """A/B switch: MFMA-fused bottleneck kernels (default) vs the generic kernels; bit-identical."""
    check(lib().ssal_set_kernel_family(1 if use_mfma else 0))


def get_knobs():
    """state of every switch that can change what a launch does (include/ssal_enet.h: ssal_debug_get_knobs)"""
    import json
    buf = ctypes.create_string_buffer(512)
    check(lib().ssal_debug_get_knobs(buf, len(buf)))
    out = json.loads(buf.value.decode())
    out["version"] = lib().ssal_version().decode()
    # the runtime environment belongs to "what a launch does" too: which values this package injected at import
    inj = sorted(set(ENV_INJECTED) | ({"GPU_MAX_HW_QUEUES"} if os.environ.get("SSAL_BENCH_INJECTED_HWQ") else set()))
    out["env"] = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "injected": inj}
    return out


def set_knob(name, value):
    """tuning / A-B knob of the fused bottleneck launchers (include/ssal_enet.h: ssal_debug_set_knob)"""
    check(lib().ssal_debug_set_knob(name.encode(), int(value)))
