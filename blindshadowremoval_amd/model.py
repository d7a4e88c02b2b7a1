"""Drop-in counterpart of the reference's ``Generator`` (/root/reference/model.py:198-290) for
inference on MI355X: same constructor, same call signature and return order, NHWC float32 tensors —
backed by libbsr_hip (hand-written gfx950 kernels) through the C ABI of include/bsr_hip.h.

    gen = Generator()
    gen.load_weights(weights)                    # dict: reference checkpoint names -> numpy arrays
    gs, con_rgb, mask22, dif = gen(inputs, uv, reg, chuck=1, training=False)

``reg`` and ``chuck`` are accepted and unused, exactly as in the reference's GSC forward
(``ShareLayer`` is constructed but never called there: model.py:221 vs :228-290).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .pack import DTYPES, pack_generator
from .tf_bundle import latest_checkpoint, load_generator_weights


class Generator:
    def __init__(self, downsize: int = 1, n_res: int = 6, device: Optional[int] = None, dtype: str = "f32"):
        """``dtype``: "f32" (fp32 matrix cores — the measured path and the default), "f32x3" (split-precision fp32 on the fp16 matrix
        cores: same end-to-end tolerance, ~1.9x faster, activations must stay below 65520 in magnitude — a violation is detected on
        the device and raised by ``check_range()`` / the next call, never silent) or "f16" (BASELINE configs[3])."""
        if dtype not in DTYPES:
            raise ValueError("dtype must be 'f32' (the measured path), 'f32x3' (split-precision fp32 on the 16-bit matrix cores) or "
                             "'f16' (fp16 operands on the 3x3-conv path, BASELINE configs[3])")
        self.dtype = dtype
        if n_res != 6:
            raise ValueError("the GSC generator has n_res=6 (/root/reference/model.py:199)")
        self.n_res = n_res
        self.n_ch = [32, 64, 64, 96, 128, 256, 256]
        self._device = device
        self._handle: Optional[ctypes.c_void_p] = None
        self._lib = None
        self._shape: Optional[Tuple[int, int, int]] = None

    # -- weights ----------------------------------------------------------------------------
    def load_weights(self, weights: Dict[str, np.ndarray]) -> "Generator":
        """``weights``: the reference's ``generator/...`` variables by checkpoint name
        (``conv1/conv/kernel`` HWIO, ConvT kernels ``[kh,kw,out,in]``, BN gamma/beta/moving_*)."""
        if not torch.cuda.is_available():
            raise RuntimeError("blindshadowremoval_amd.Generator needs a ROCm GPU: there is no CPU path")
        lib = _lib.load()
        dev = torch.cuda.current_device() if self._device is None else int(self._device)
        blob = pack_generator(weights, self.dtype)
        handle = ctypes.c_void_p()
        buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
        with torch.cuda.device(dev):
            rc = lib.bsr_create(ctypes.byref(handle), dev, ctypes.cast(buf, ctypes.c_void_p), len(blob), DTYPES[self.dtype])
        _lib.check(rc, "bsr_create")
        self.close()
        self._lib, self._handle, self._device = lib, handle, dev
        return self

    def restore(self, checkpoint_dir: str) -> int:
        """Counterpart of ``tf.train.latest_checkpoint`` + ``checkpoint.restore(...).expect_partial()``
        (/root/reference/train_test_GSC.py:362-367).  Returns the epoch parsed from the file name, 0 if none."""
        prefix = latest_checkpoint(checkpoint_dir)
        if not prefix:
            return 0
        self.load_weights(load_generator_weights(prefix))
        return int(prefix.split("-")[-1])

    def close(self) -> None:
        if self._handle is not None and self._lib is not None:
            self._lib.bsr_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- forward ----------------------------------------------------------------------------
    def _check_input(self, t: torch.Tensor, name: str, dev: int) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        if t.dim() != 4 or t.shape[-1] != 3:
            raise ValueError("%s must be [B,H,W,3] NHWC, got %s" % (name, tuple(t.shape)))
        if t.dtype != torch.float32:
            raise TypeError("%s must be float32, got %s" % (name, t.dtype))
        if t.device.type != "cuda" or t.device.index != dev:
            t = t.to("cuda:%d" % dev)
        return t.contiguous()

    @staticmethod
    def _check_out(out, B: int, H: int, W: int, dev: int):
        """Caller-supplied output buffers go to the library as raw pointers: refuse anything it would write out of bounds."""
        if not isinstance(out, (tuple, list)) or len(out) != 4:
            raise ValueError("out must be a tuple (gs, con_rgb, mask22, dif)")
        for t, c, name in zip(out, (1, 3, 3, 1), ("gs", "con_rgb", "mask22", "dif")):
            if not isinstance(t, torch.Tensor) or tuple(t.shape) != (B, H, W, c):
                raise ValueError("out[%s] must be a tensor of shape %s, got %s" % (name, (B, H, W, c), tuple(getattr(t, "shape", ()))))
            if t.dtype != torch.float32:
                raise TypeError("out[%s] must be float32, got %s" % (name, t.dtype))
            if t.device.type != "cuda" or t.device.index != dev:
                raise ValueError("out[%s] must live on cuda:%d, got %s" % (name, dev, t.device))
            if not t.is_contiguous():
                raise ValueError("out[%s] must be contiguous (dense NHWC)" % name)
        return tuple(out)

    def __call__(self, inputs, uv, reg=None, chuck: int = 1, training: bool = False,
                 out: Optional[Tuple[torch.Tensor, ...]] = None, packed_out: Optional[torch.Tensor] = None):
        """``packed_out``: optional [B,H,W,4] float32 CUDA tensor; con_rgb | dif are then written straight into it (bsr_forward_packed:
        the multi-GPU all-gather payload) and the returned con_rgb / dif are views of it."""
        if training:
            raise NotImplementedError("only the inference path (training=False) is implemented "
                                      "(/root/reference/train_test_GSC.py:404,856)")
        if self._handle is None:
            raise RuntimeError("Generator has no weights: call load_weights() or restore() first")
        dev = self._device
        inputs = self._check_input(inputs, "inputs", dev)
        uv = self._check_input(uv, "uv", dev)
        if uv.shape != inputs.shape:
            raise ValueError("inputs %s and uv %s must have the same shape" % (tuple(inputs.shape), tuple(uv.shape)))
        B, H, W, _ = inputs.shape
        if H % 32 or W % 256:
            raise ValueError("H must be a multiple of 32 and W of 256 (reference IMG_SIZE = 256), got %dx%d" % (H, W))
        with torch.cuda.device(dev):
            if out is None:
                gs = torch.empty((B, H, W, 1), dtype=torch.float32, device=inputs.device)
                con_rgb = torch.empty((B, H, W, 3), dtype=torch.float32, device=inputs.device)
                mask22 = torch.empty((B, H, W, 3), dtype=torch.float32, device=inputs.device)
                dif = torch.empty((B, H, W, 1), dtype=torch.float32, device=inputs.device)
            else:
                gs, con_rgb, mask22, dif = self._check_out(out, B, H, W, dev)
            stream = torch.cuda.current_stream().cuda_stream
            if packed_out is not None:
                if (not isinstance(packed_out, torch.Tensor) or tuple(packed_out.shape) != (B, H, W, 4) or packed_out.dtype != torch.float32
                        or packed_out.device.type != "cuda" or packed_out.device.index != dev or not packed_out.is_contiguous()):
                    raise ValueError("packed_out must be a contiguous float32 [B,H,W,4] tensor on cuda:%d" % dev)
                rc = self._lib.bsr_forward_packed(self._handle, inputs.data_ptr(), uv.data_ptr(), B, H, W, gs.data_ptr(), packed_out.data_ptr(),
                                                  mask22.data_ptr(), stream)
                con_rgb, dif = packed_out[..., :3], packed_out[..., 3:]
            else:
                rc = self._lib.bsr_forward(self._handle, inputs.data_ptr(), uv.data_ptr(), B, H, W, gs.data_ptr(), con_rgb.data_ptr(),
                                           mask22.data_ptr(), dif.data_ptr(), stream)
        _lib.check(rc, "bsr_forward")
        self._shape = (B, H, W)
        return gs, con_rgb, mask22, dif

    # -- TSM variant ------------------------------------------------------------------------
    def call_tsm(self, inputs, uv, reg, frame: int, share: bool = True, chuck: int = 1, training: bool = False):
        """``Generator.call(inputs, uv, reg, frame, share, chuck, training)`` of /root/reference/model_with_TSM.py:261-325
        (call site /root/reference/train_with_TSM.py:676).  Needs TSM weights (291-channel ``res_stack/0/conv1``)."""
        if training:
            raise NotImplementedError("only the inference path (training=False) is implemented")
        if self._handle is None:
            raise RuntimeError("Generator has no weights: call load_weights() or restore() first")
        dev = self._device
        inputs = self._check_input(inputs, "inputs", dev)
        uv = self._check_input(uv, "uv", dev)
        if not isinstance(reg, torch.Tensor):
            reg = torch.as_tensor(np.asarray(reg))
        if reg.dim() != 4 or reg.shape[-1] != 6 or reg.shape[:3] != inputs.shape[:3] or reg.dtype != torch.float32:
            raise ValueError("reg must be float32 [B,H,W,6] (reg_in | reg_out), got %s %s" % (tuple(reg.shape), reg.dtype))
        reg = reg.to("cuda:%d" % dev).contiguous()
        B, H, W, _ = inputs.shape
        if H != W or H % 256:
            raise ValueError("the TSM path needs square images with H a multiple of 256, got %dx%d" % (H, W))
        if frame <= 0 or B % frame:
            raise ValueError("batch %d is not a multiple of frame %d" % (B, frame))
        with torch.cuda.device(dev):
            gs = torch.empty((B, H, W, 1), dtype=torch.float32, device=inputs.device)
            con_rgb = torch.empty((B, H, W, 3), dtype=torch.float32, device=inputs.device)
            mask22 = torch.empty((B, H, W, 3), dtype=torch.float32, device=inputs.device)
            dif = torch.empty((B, H, W, 1), dtype=torch.float32, device=inputs.device)
            stream = torch.cuda.current_stream().cuda_stream
            rc = self._lib.bsr_forward_tsm(self._handle, inputs.data_ptr(), uv.data_ptr(), reg.data_ptr(), B, H, W, int(frame), 1 if share else 0,
                                           gs.data_ptr(), con_rgb.data_ptr(), mask22.data_ptr(), dif.data_ptr(), stream)
        _lib.check(rc, "bsr_forward_tsm")
        self._shape = (B, H, W)
        return gs, con_rgb, mask22, dif

    def check_range(self) -> None:
        """16-bit modes (dtype "f32x3" / "f16"): synchronise the current stream and raise ``_lib.RangeError`` if any forward since
        the last call converted an activation of magnitude >= 65520 to fp16 (its outputs hold inf / NaN where the fp32 path stays
        finite: discard them and re-run on a ``dtype="f32"`` generator).  The kernels detect this on the device (bsr_check_range in
        include/bsr_hip.h); without a call the NEXT forward after a completed overflowing one raises instead.  No-op for "f32"."""
        if self._handle is None:
            raise RuntimeError("Generator has no weights: call load_weights() or restore() first")
        with torch.cuda.device(self._device):
            rc = self._lib.bsr_check_range(self._handle, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "bsr_check_range")

    def peek_range(self) -> None:
        """``check_range`` for a pipelined caller: no stream synchronisation and the condition is not cleared — raises
        ``_lib.RangeError`` if any forward that has COMPLETED so far overflowed fp16 (bsr_peek_range).  No-op for "f32"."""
        if self._handle is None:
            raise RuntimeError("Generator has no weights: call load_weights() or restore() first")
        _lib.check(self._lib.bsr_peek_range(self._handle), "bsr_peek_range")

    # -- test / measurement hooks -----------------------------------------------------------
    def probe(self, name: str) -> torch.Tensor:
        """Intermediate of the last forward as a dense NHWC tensor (see bsr_probe in include/bsr_hip.h)."""
        if self._shape is None:
            raise RuntimeError("probe() needs a forward first")
        shape = (ctypes.c_int * 4)()
        with torch.cuda.device(self._device):
            stream = torch.cuda.current_stream().cuda_stream
            # first call with zero capacity only reports the shape (the copy is refused with BSR_ERR_ARG) ...
            rc = self._lib.bsr_probe(self._handle, name.encode(), ctypes.c_void_p(8), 0, ctypes.byref(shape), stream)
            if rc not in (0, 1) or shape[0] == 0:
                _lib.check(rc if rc else 4, "bsr_probe")
            n = shape[0] * shape[1] * shape[2] * shape[3]
            dst = torch.empty(n, dtype=torch.float32, device="cuda:%d" % self._device)
            # ... the second one copies into an exactly sized tensor
            _lib.check(self._lib.bsr_probe(self._handle, name.encode(), dst.data_ptr(), dst.numel(), ctypes.byref(shape), stream), "bsr_probe")
        return dst.reshape(shape[0], shape[1], shape[2], shape[3])

    def set_timing(self, enable: bool) -> None:
        _lib.check(self._lib.bsr_set_timing(self._handle, 1 if enable else 0), "bsr_set_timing")

    def get_timing(self) -> Dict[str, Tuple[float, int]]:
        ms = (ctypes.c_float * _lib.NUM_CLASSES)()
        n = (ctypes.c_int * _lib.NUM_CLASSES)()
        _lib.check(self._lib.bsr_get_timing(self._handle, ctypes.byref(ms), ctypes.byref(n)), "bsr_get_timing")
        return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(_lib.CLASS_NAMES)}


    def get_launch_timing(self):
        """[(layer name, ms, class name)] of the last timed forward, in launch order (bsr_timing_entry)."""
        out = []
        name = ctypes.create_string_buffer(64)
        ms, cls = ctypes.c_float(), ctypes.c_int()
        for i in range(self._lib.bsr_timing_launches(self._handle)):
            _lib.check(self._lib.bsr_timing_entry(self._handle, i, name, 64, ctypes.byref(ms), ctypes.byref(cls)), "bsr_timing_entry")
            out.append((name.value.decode(), float(ms.value), _lib.CLASS_NAMES[cls.value]))
        return out


class GeneratorTSM(Generator):
    """Drop-in for ``Generator`` of /root/reference/model_with_TSM.py:231-325 (temporal-sharing variant, BASELINE config 5):
    ``gen(inputs, uv, reg, frame, share, chuck, training)``."""

    def __call__(self, inputs, uv, reg, frame, share=True, chuck: int = 1, training: bool = False):
        return self.call_tsm(inputs, uv, reg, frame, share, chuck, training)
