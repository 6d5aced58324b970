"""ctypes binding of libsparsify_hip.so (the C ABI declared in include/sparsify_hip.h).

The prototypes are read from the header itself, so the binding cannot drift from the ABI.  There is no CPU
fallback anywhere in this package: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "sparsify_hip.h")
LIB_PATH = os.path.join(HERE, "libsparsify_hip.so")

SC_F32, SC_BF16 = 0, 1


class ScError(RuntimeError):
    pass


class GemmEpilogue(ctypes.Structure):
    _fields_ = [("alpha", ctypes.c_float), ("beta", ctypes.c_float), ("bias", ctypes.c_void_p), ("pre_out", ctypes.c_void_p),
                ("act", ctypes.c_int32), ("resid_dtype", ctypes.c_int32), ("resid", ctypes.c_void_p),
                ("dgelu_pre", ctypes.c_void_p), ("ld_aux", ctypes.c_int64),
                ("colsum", ctypes.c_void_p), ("colsum_ws", ctypes.c_void_p), ("colsum_ws_bytes", ctypes.c_uint64),
                ("colsum_accumulate", ctypes.c_int32), ("reserved_", ctypes.c_int32), ("tile_tickets", ctypes.c_void_p)]


_BLOCK_PTRS_1 = ["ln1_g", "ln1_b", "b_qkv", "b_o", "ln2_g", "ln2_b", "b_fc1", "b_fc2",
                 "w_qkv", "w_o", "w_fc1", "w_fc2", "wt_qkv", "wt_o", "wt_fc1", "wt_fc2",
                 "x_in", "x_mid", "x_out", "ln1_out", "qkv", "attn_out", "ln2_out", "h_pre", "h_act",
                 "ln1_mean", "ln1_rstd", "ln2_mean", "ln2_rstd",
                 "g_ln1_g", "g_ln1_b", "g_w_qkv", "g_b_qkv", "g_w_o", "g_b_o", "g_ln2_g", "g_ln2_b", "g_w_fc1", "g_b_fc1",
                 "g_w_fc2", "g_b_fc2"]
_BLOCK_PTRS_2 = ["d_h", "d_ln", "d_qkv", "d_attn", "d_res_t", "dx_mid", "ws"]


class BlockDesc(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int64) for n in ["batch", "seq", "width", "heads", "mlp_width"]] +
                [("dtype", ctypes.c_int32), ("causal", ctypes.c_int32)] +
                [(n, ctypes.c_void_p) for n in _BLOCK_PTRS_1] +
                [("accumulate", ctypes.c_int32), ("b_fc2_done", ctypes.c_int32)] +
                [(n, ctypes.c_void_p) for n in _BLOCK_PTRS_2] +
                [("ws_bytes", ctypes.c_size_t), ("g_below_b_fc2", ctypes.c_void_p), ("ws_side", ctypes.c_void_p), ("ws_side_bytes", ctypes.c_size_t),
                 ("events", ctypes.c_void_p * 4), ("tile_tickets", ctypes.c_void_p), ("attn_lse", ctypes.c_void_p)])


_CTYPES = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
           "size_t": ctypes.c_size_t, "uint8_t": ctypes.c_uint8}


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes])} for every function prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"\b(const char\s*\*|size_t|int)\s+(sc_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else _CTYPES[ret.strip()]
        argtypes = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[a.split()[-2] if len(a.split()) > 1 else a])
        protos[name] = (restype, argtypes)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise ScError(f"{LIB_PATH} is missing: build it with `python -m sparsify_clip_amd._build` "
                              "(there is no CPU fallback)")
            dll = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes) in self.protos.items():
                fn = getattr(dll, name)  # AttributeError if the library does not export a declared symbol
                fn.restype = restype
                fn.argtypes = argtypes
            self._dll = dll
        return self._dll

    def raw(self, name):
        return getattr(self.load(), name)

    def call(self, name, *args):
        """Call an int-returning entry point; non-zero status raises with sc_last_error()."""
        rc = getattr(self.load(), name)(*args)
        if rc != 0:
            msg = self._dll.sc_last_error()
            raise ScError(f"{name} failed with status {rc}: {msg.decode() if msg else ''}")


LIB = _Lib()


class EventSet:
    """Four caller-owned HIP events for sc_block_desc.events (sc_block_bwd_async orders its two streams with them)."""

    def __init__(self):
        self.handles = []
        for _ in range(4):
            h = ctypes.c_void_p()
            LIB.call("sc_event_create", ctypes.byref(h))
            self.handles.append(h.value)

    def bind(self, desc):
        for i, h in enumerate(self.handles):
            desc.events[i] = h

    def __del__(self):
        try:
            for h in self.handles:
                LIB.raw("sc_event_destroy")(ctypes.c_void_p(h))
        except Exception:
            pass
        self.handles = []


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    """The current PyTorch-ROCm HIP stream as a void* (all library work is enqueued on it)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t: torch.Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise ScError(f"{name} must live on the GPU: this package has no CPU path")
    if not t.is_contiguous():
        raise ScError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise ScError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def sc_dtype(t: torch.dtype) -> int:
    if t == torch.float32:
        return SC_F32
    if t == torch.bfloat16:
        return SC_BF16
    raise ScError(f"unsupported dtype {t}")
