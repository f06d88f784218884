"""Device-side state of one MultiMNIST MMVAE instance and the fused ELBO-step engine.

PyTorch is used for plumbing only: it owns the device allocations (``torch.empty``), the stream and
(optionally) the HIP graph; every arithmetic operation of the hot path runs in libmmvae_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import MMVAEError, StepIO, call, ptr


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class PlanState:
    """Flat parameter / gradient / BatchNorm-buffer storage plus the packed bf16 GEMM copies of the weights.

    The nn.Module face (``multimnist.MultimodalVAE`` ...) exposes views of ``params`` as its nn.Parameters, so
    ``optim.Adam(vae.parameters())``, ``state_dict()`` and the fused engine all see the same memory.
    ``API`` is the C-ABI prefix of the model family (mmvae_<API>_create ...)."""

    API = "mm"

    def _c(self, fn, *args):
        return call("mmvae_%s_%s" % (self.API, fn), *args)

    def __init__(self, n_latents: int, device: torch.device):
        if device.type != "cuda":
            raise MMVAEError("the MMVAE HIP engine needs a gfx950 GPU (device=%s); there is no CPU fallback" % device)
        _lib.init_device(device.index if device.index is not None else torch.cuda.current_device())
        self.device = device
        self.n_latents = int(n_latents)
        self._plans: Dict[int, int] = {}
        h = self._handle(1, bind=False)
        self.nparams = self._c("param_count", h)
        self.table: List[Tuple[str, Tuple[int, ...], int]] = []
        name = C.create_string_buffer(128)
        nd, off = C.c_int(), C.c_longlong()
        shape = (C.c_int * 4)()
        for i in range(self._c("num_params", h)):
            self._c("param_info", h, i, name, C.byref(nd), shape, C.byref(off))
            self.table.append((name.value.decode(), tuple(shape[k] for k in range(nd.value)), off.value))
        self.bn_table: List[Tuple[str, int, int]] = []
        ch = C.c_int()
        for i in range(self._c("num_bn", h)):
            self._c("bn_info", h, i, name, C.byref(ch), C.byref(off))
            self.bn_table.append((name.value.decode(), ch.value, off.value))
        f32 = dict(dtype=torch.float32, device=device)
        self.params = torch.zeros(self.nparams, **f32)
        self.grads = torch.zeros(self.nparams, **f32)
        self.bn_stats = torch.zeros(self._c("bn_floats", h), **f32)
        for _, c, o in self.bn_table:
            self.bn_stats[o + c:o + 2 * c] = 1.0                       # running_var = 1
        self.bn_nbt = torch.zeros(len(self.bn_table), dtype=torch.int64, device=device)
        self.packed = torch.zeros(self._c("packed_elems", h), dtype=torch.bfloat16, device=device)
        self.packed_vec = torch.zeros(self._c("packed_vec_elems", h), **f32)
        self.gpk = torch.zeros(self._c("gpk_elems", h), **f32)
        self.gpk_vec = torch.zeros(self._c("gpk_vec_elems", h), **f32)
        self._desc = []
        for which in (0, 1):
            nbytes = self._c("desc_bytes", h, which)
            host = torch.empty(nbytes, dtype=torch.uint8)
            self._c("desc_copy", h, which, C.c_void_p(host.data_ptr()))
            self._desc.append(host.to(device))
        self._bind(h)
        self.packed_version = -1          # bumped by whoever changes params

    # ---- plans -------------------------------------------------------------------------------------
    def _handle(self, batch: int, bind: bool = True) -> int:
        h = self._plans.get(batch)
        if h is None:
            h = self._c("create", self.n_latents, int(batch))
            if not h:
                raise MMVAEError("mmvae_%s_create failed: %s" % (self.API, _lib.load().mmvae_last_error().decode()))
            self._plans[batch] = h
            if bind:
                self._bind(h)
        return h

    def _bind(self, h: int) -> None:
        self._c("bind", h, ptr(self.params), ptr(self.grads), ptr(self.bn_stats), ptr(self.bn_nbt),
             ptr(self.packed), ptr(self.packed_vec), ptr(self.gpk), ptr(self.gpk_vec), ptr(self._desc[0]), ptr(self._desc[1]))

    def rebind(self) -> None:
        """Re-point every plan at the current buffers (after the module moved its flat storage)."""
        for h in self._plans.values():
            self._bind(h)

    def plan(self, batch: int) -> int:
        return self._handle(int(batch))

    def workspace_bytes(self, batch: int) -> int:
        return self._c("workspace_bytes", self.plan(batch))

    def module_workspace_bytes(self, batch: int) -> int:
        """Workspace of one granular module call (one pass over ``batch`` rows), not of the whole 3-pass step."""
        return self._c("module_workspace_bytes", self.plan(batch))

    def grad_map(self) -> torch.Tensor:
        """int32 [nparams]: where each flat parameter's packed gradient lives (mmvae_<family>_grad_map), built once."""
        if getattr(self, "_gmap", None) is None:
            m = torch.empty(self.nparams, dtype=torch.int32, device=self.device)
            self._c("grad_map", self.plan(1), ptr(m), _stream())
            self._gmap = m
        return self._gmap

    def pack_weights(self) -> None:
        self._c("pack_weights", self.plan(1), _stream())
        self.pack_pending = False

    pack_pending = False        # the parameters changed and the bf16 GEMM copies were not refreshed yet

    def ensure_packed(self) -> None:
        """Refresh the packed weights if an optimizer step deferred it (the MultiMNIST fused step folds the refresh into
        its own prologue launch; everybody else who reads the packed weights calls this first)."""
        if self.pack_pending:
            self.pack_weights()

    def view(self, name: str) -> torch.Tensor:
        for n, shape, off in self.table:
            if n == name:
                numel = 1
                for s in shape:
                    numel *= s
                return self.params[off:off + numel].view(shape)
        raise KeyError(name)

    def __del__(self):
        try:
            for h in self._plans.values():
                self._c("destroy", h)
        except Exception:
            pass


class MultimnistState(PlanState):
    """multimnist/model.py MultimodalVAE"""
    API = "mm"


class MnistState(PlanState):
    """mnist/model.py MultimodalVAE.  ``precision``: "fp32" (default: the reference's own arithmetic on fp32 MFMA) or
    "bf16" (bf16 MFMA operands like the conv models)."""
    API = "mnist"

    def __init__(self, n_latents: int, device: torch.device, precision: str = "fp32"):
        assert precision in ("fp32", "bf16")
        self.precision = precision
        super().__init__(n_latents, device)

    def _c(self, fn, *args):
        if fn == "create":
            return call("mmvae_mnist_create_p", *args, 0 if self.precision == "fp32" else 1)
        return super()._c(fn, *args)


class CelebaState(PlanState):
    """celeba/model.py MultimodalVAE"""
    API = "celeba"


class CocoState(PlanState):
    """coco/model.py MultimodalVAE.  ``steps``: caption length (coco/utils.py:12-15: 102)."""
    API = "coco"

    def __init__(self, n_latents: int, device: torch.device, steps: int = 102):
        self.steps = int(steps)
        super().__init__(n_latents, device)

    def _c(self, fn, *args):
        if fn == "create":
            return call("mmvae_coco_create_t", *args, self.steps)
        return super()._c(fn, *args)


class StepOutputs:
    """Lazy view of the loss sums of one fused step (no host sync until a value is read)."""

    def __init__(self, sums: torch.Tensor, bce_div: float, nll_div: float, kl_scale: float, lambda_xy, lambda_yx,
                 passes=(True, True, True)):
        self.sums, self.bce_div, self.nll_div, self.kl_scale = sums, bce_div, nll_div, kl_scale
        self.lxy, self.lyx = tuple(float(x) for x in lambda_xy), tuple(float(x) for x in lambda_yx)
        self.passes = tuple(bool(x) for x in passes)

    def losses(self) -> torch.Tensor:
        """loss_1, loss_2, loss_3 of the reference's train() closure as a device tensor [3]: the weighted sums that end each
        loss_function call (multimnist/train.py:33-62), evaluated by the library (mmvae_step_losses) on the step's stream --
        the weights travel as kernel arguments, nothing is synchronised."""
        s = self.sums
        w = [(C.c_float * 3)() for _ in range(3)]
        for k in range(3):
            on = 1.0 if self.passes[k] else 0.0                   # absent passes (weak supervision) report 0
            w[0][k] = on * self.lxy[k] / self.bce_div
            w[1][k] = on * self.lyx[k] / self.nll_div
            w[2][k] = on * self.kl_scale
        out = torch.empty(3, dtype=torch.float32, device=s.device)
        call("mmvae_step_losses", ptr(s), w[0], w[1], w[2], ptr(out), _stream())
        return out

    def check(self) -> None:
        """Synchronises, then raises MMVAEError if the step declared itself void (mmvae_step_status: a device-side exchange of
        the COCO caption decoder gave up; the optimizer update of that step was skipped on every rank)."""
        call("mmvae_step_status", ptr(self.sums), _stream())

    def parts(self):
        """(mean BCE, mean NLL, KL sum) per pass"""
        s = self.sums
        return s[0:3] / self.bce_div, s[4:7] / self.nll_div, s[8:11]


class _FusedStepBase:
    """zero_grad -> 3-pass forward -> 3 losses -> backward -> [grad all-reduce] -> Adam, as ONE enqueue of HIP kernels."""

    def __init__(self, state: PlanState, batch: int, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 seed: int = 1234, world_size: int = 1, all_reduce=None):
        self.state, self.B, self.D = state, int(batch), state.n_latents
        self.lr, self.betas, self.eps, self.seed = lr, betas, eps, seed
        self.world_size, self.all_reduce = world_size, all_reduce
        if all_reduce is not None and (world_size > 1 or getattr(all_reduce, "force", False)):
            # the collective library may enqueue on a stream of its own: mixed stream priorities then slow the whole
            # process down (include/mmvae_hip.h: mmvae_set_stream_policy)
            call("mmvae_set_stream_policy", 1)
        dev = state.device
        self.h = state.plan(batch)
        self.ws = torch.empty(state.workspace_bytes(batch), dtype=torch.uint8, device=dev)
        self.exp_avg = torch.zeros_like(state.params)
        self.exp_avg_sq = torch.zeros_like(state.params)
        self.adam_state = torch.zeros(2, dtype=torch.int64, device=dev)      # {step, ticket}
        self.sums = torch.zeros(16, dtype=torch.float32, device=dev)
        self._graph = None
        state.pack_weights()

    def _outputs(self) -> StepOutputs:
        raise NotImplementedError

    def _pass_config(self, io, passes, la, lb, name_a, name_b):
        """Fills the lambda arrays and pass_skip flags of a step-io struct; returns (passes, lambda_a, lambda_b)."""
        passes = (True, True, True) if passes is None else tuple(bool(x) for x in passes)
        assert len(passes) == 3 and any(passes), "at least one of the three passes must be present"
        la = tuple(float(x) for x in (self._LA if la is None else la))
        lb = tuple(float(x) for x in (self._LB if lb is None else lb))
        setattr(io, name_a, (C.c_float * 3)(*la))
        setattr(io, name_b, (C.c_float * 3)(*lb))
        io.pass_skip = (C.c_int * 3)(*[0 if x else 1 for x in passes])
        self._last = (passes, la, lb)
        return passes, la, lb

    def optimizer_step(self) -> None:
        """torch.optim.Adam(lr) semantics on the flat buffers, then refresh the packed bf16 weights."""
        st = self.state
        if self._dp_active():
            self.all_reduce(st.grads)                                  # sum over ranks (RCCL), scaled inside Adam
        call("mmvae_adam_step", ptr(st.params), ptr(st.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), st.nparams,
             ptr(self.adam_state), self.lr, self.betas[0], self.betas[1], self.eps, 1.0 / self.world_size, _stream())
        self._after_update()

    _DEFER_PACK = False         # the family's step prologue can refresh the packed weights itself (pack_first)
    separate_unpack = False     # tests: unpack the gradient + plain Adam instead of the packed-gradient Adam (same numbers)

    def _after_update(self) -> None:
        if self._DEFER_PACK:
            self.state.pack_pending = True
        else:
            self.state.pack_weights()

    def _dp_active(self) -> bool:
        """The data-parallel exchange is on: more than one rank, or a collective that asks to run even on one rank
        (``all_reduce.force``: the single-rank RCCL test that puts this branch on hardware)."""
        return self.all_reduce is not None and (self.world_size > 1 or bool(getattr(self.all_reduce, "force", False)))

    def __call__(self, a, b, **kw) -> StepOutputs:
        out = self.forward_backward(a, b, True, True, **kw)
        self.optimizer_step()
        return out

    def _call_packed(self, a, b, **kw) -> StepOutputs:
        """backward + optimizer.step() (multimnist/train.py:168,173) in one pass over the parameters: the GEMM-weight
        gradients stay in their packed layout and the Adam kernel gathers them (and completes ``grads``) itself.  The
        data-parallel path needs the complete flat gradient BEFORE Adam (all-reduce), so it keeps the separate unpack."""
        if self._dp_active() or self.separate_unpack:
            return _FusedStepBase.__call__(self, a, b, **kw)
        st = self.state
        out = self.forward_backward(a, b, True, True, _defer_unpack=True, **kw)
        call("mmvae_adam_step_packed", ptr(st.params), ptr(st.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), st.nparams,
             ptr(self.adam_state), self.lr, self.betas[0], self.betas[1], self.eps, 1.0, ptr(st.grad_map()), ptr(st.gpk),
             ptr(st.gpk_vec), _stream())
        self._after_update()
        return out

    # -- HIP graph ------------------------------------------------------------------------------------
    def capture(self, a: torch.Tensor, b: torch.Tensor) -> None:
        """Capture the whole step (static input buffers) into a HIP graph; ``replay()`` launches it."""
        self._static = (a, b)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):          # warm-up: sets kernel attributes, primes the allocator
                self(a, b)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self(a, b)
        self._graph = g

    def replay(self) -> StepOutputs:
        self._graph.replay()
        return self._outputs()


class FusedELBOStep(_FusedStepBase):
    """The 3-pass ELBO step of multimnist/train.py:146-173."""

    LAMBDA_XY = (1.0, 1.0, 0.0)
    LAMBDA_YX = (1.0, 0.5, 1.0)
    _DEFER_PACK = True

    def __init__(self, state: PlanState, batch: int, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 kl_lambda: float = 1e-3, seed: int = 1234, world_size: int = 1, all_reduce=None):
        super().__init__(state, batch, lr, betas, eps, seed, world_size, all_reduce)
        self.kl_lambda = kl_lambda
        self.enc_dropout = self.gru_dropout = True
        self._LA, self._LB = self.LAMBDA_XY, self.LAMBDA_YX
        self._last = ((True, True, True), self.LAMBDA_XY, self.LAMBDA_YX)

    def _outputs(self) -> StepOutputs:
        passes, lxy, lyx = self._last
        return StepOutputs(self.sums, self.B * 2500, self.B * 4, self.kl_lambda / self.B, lxy, lyx, passes)

    def forward_backward(self, image, text, training=True, backward=True, eps=None, enc_mask1=None, enc_mask2=None,
                         gru_keep=None, force_tokens=None, recon_image=None, recon_text=None, mu=None, logvar=None,
                         tokens=None, passes=None, lambda_xy=None, lambda_yx=None, _defer_unpack=False, _dp_split=False,
                         _early_adam=None) -> StepOutputs:
        """``passes`` = which of (joint, image-only, text-only) exist in this step and ``lambda_xy/lambda_yx`` = their loss
        weights: the weak-supervision steps of multimnist/paired_weak.py:84-117 and modal_weak.py:87-117."""
        assert image.is_contiguous() and text.is_contiguous() and image.dtype == torch.float32 and text.dtype == torch.int64
        assert image.shape[0] == self.B and text.shape == (self.B, 4)
        io = StepIO()
        io.ws, io.ws_bytes = self.ws.data_ptr(), self.ws.numel()
        io.step_counter = self.adam_state.data_ptr()
        io.image, io.text = image.data_ptr(), text.data_ptr()
        for k, t in (("eps", eps), ("enc_mask1", enc_mask1), ("enc_mask2", enc_mask2), ("gru_keep", gru_keep),
                     ("force_tokens", force_tokens), ("recon_image", recon_image), ("recon_text", recon_text),
                     ("mu", mu), ("logvar", logvar), ("tokens", tokens)):
            setattr(io, k, None if t is None else t.data_ptr())
        io.enc_dropout, io.gru_dropout = int(self.enc_dropout), int(self.gru_dropout)
        io.kl_lambda = self.kl_lambda
        self._pass_config(io, passes, lambda_xy, lambda_yx, "lambda_xy", "lambda_yx")
        io.seed = self.seed
        io.sums = self.sums.data_ptr()
        io.defer_unpack = int(bool(_defer_unpack))
        io.pack_first = int(self.state.pack_pending)
        io.dp_split = int(bool(_dp_split))
        io.early_adam = _early_adam
        # optimizer.zero_grad() (train.py:150) happens inside the step's prologue kernel when backward is requested
        call("mmvae_mm_step", self.h, C.byref(io), int(training), int(backward), _stream())
        self.state.pack_pending = False
        return self._outputs()

    # -- data parallelism: the decoders' gradients are exchanged while the encoders' backward still runs -------------
    EARLY_PREFIXES = ("image_decoder.", "text_decoder.")     # what mmvae_mm_step_io.dp_split completes early

    def grad_ranges(self):
        """(early, late): lists of (offset, length) runs of the flat gradient buffer, early = complete at the event of
        ``mmvae_mm_wait_early_grads``.  Adjacent parameters of the same kind are merged: a few large messages."""
        runs = ([], [])
        end = 0
        for i, (name, shape, off) in enumerate(self.state.table):
            n = 1
            for d in shape:
                n *= int(d)
            kind = 0 if name.startswith(self.EARLY_PREFIXES) else 1
            r = runs[kind]
            if r and r[-1][0] + r[-1][1] == off and getattr(self, "_last_kind", None) == kind:
                r[-1] = (r[-1][0], r[-1][1] + n)
            else:
                r.append((off, n))
            self._last_kind = kind
            end = max(end, off + n)
        assert end == self.state.nparams
        return runs

    def _call_dp_overlap(self, image, text, **kw) -> StepOutputs:
        """One data-parallel step with the gradient exchange in two parts: the early ranges are all-reduced on a
        communication stream ordered behind the step's early-gradient event (they overlap latent/encoder backward and the
        encoders' weight gradients), the late ranges after the step; Adam (1/world folded in) waits for both.

        Opt-in (``GradAllReduce(overlap=True)``), correct (tests/test_gpu_configs_r2.py) but NOT faster on this runtime: a
        stream parked on an event occupies its hardware queue, and when the runtime maps the communication stream onto a
        queue shared with one of the step's streams the step stalls behind it (0.91 -> 2.2 ms measured with a world-1 RCCL
        group, GPU_MAX_HW_QUEUES=8).  The plain exchange after the step is the default."""
        st = self.state
        if getattr(self, "_ranges", None) is None:
            self._ranges = self.grad_ranges()
            self._comm_owner = _lib.OwnedStream(st.device)      # (not torch.cuda.Stream(): see _lib.OwnedStream)
            self._comm = self._comm_owner.stream
        early, late = self._ranges
        out = self.forward_backward(image, text, True, True, _dp_split=True, **kw)
        # async collectives: the library's stream is ordered behind the stream that is current at the call (the
        # communication stream, whose only content is the wait for the early-gradient event, for the early ranges; the main
        # stream for the late ones) and Work.wait() orders the main stream behind the results.  NOT main.wait_stream(comm):
        # an event recorded on a stream that holds nothing but a pending wait costs ~1.3 ms per step on this runtime.
        works = []
        call("mmvae_mm_wait_early_grads", self.h, C.c_void_p(self._comm.cuda_stream))
        with torch.cuda.stream(self._comm):
            for off, n in early:
                works.append(self.all_reduce(st.grads[off:off + n], async_op=True))
        for off, n in late:
            works.append(self.all_reduce(st.grads[off:off + n], async_op=True))
        for w in works:
            if w is not None:
                w.wait()
        call("mmvae_adam_step", ptr(st.params), ptr(st.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), st.nparams,
             ptr(self.adam_state), self.lr, self.betas[0], self.betas[1], self.eps, 1.0 / self.world_size, _stream())
        self._after_update()
        return out

    early_adam = True       # the decoders' Adam update is issued inside the step, beside the encoders' backward (mmvae_early_adam)

    def _early_setup(self):
        """ctypes block handed to mmvae_mm_step_io.early_adam + the ranges the call behind the step still has to update."""
        st = self.state
        rg = (C.c_longlong * 8)()
        n = call("mmvae_mm_early_ranges", self.h, rg, 4)
        early = [(int(rg[2 * i]), int(rg[2 * i + 1])) for i in range(n)]
        late, pos = [], 0
        for off, ln in sorted(early):
            if off > pos:
                late.append((pos, off - pos))
            pos = off + ln
        if pos < st.nparams:
            late.append((pos, st.nparams - pos))
        self._ea_ran = C.c_int(0)
        ea = _lib.EarlyAdam()
        ea.m, ea.v, ea.state = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.adam_state.data_ptr()
        ea.lr, ea.beta1, ea.beta2, ea.eps, ea.grad_scale = self.lr, self.betas[0], self.betas[1], self.eps, 1.0
        ea.gmap = st.grad_map().data_ptr()
        ea.ran = C.pointer(self._ea_ran)
        self._ea = ea
        self._ea_late = (C.c_longlong * (2 * len(late)))(*[x for r in late for x in r])
        self._ea_nlate = len(late)
        self._ea_ok = 0 < n and 0 < len(late) <= 4 and all(o % 4 == 0 and l % 4 == 0 for o, l in late[:-1]) and late[-1][0] % 4 == 0

    def _call_packed(self, image, text, **kw) -> StepOutputs:
        """backward + optimizer.step() with the decoders' part of the update inside the step (multimnist/train.py:168,173)."""
        if self._dp_active() or self.separate_unpack or not self.early_adam:
            return _FusedStepBase._call_packed(self, image, text, **kw)
        if getattr(self, "_ea", None) is None:
            self._early_setup()
        if not self._ea_ok:
            return _FusedStepBase._call_packed(self, image, text, **kw)
        st = self.state
        self._ea.lr = self.lr
        if not kw:
            return self._call_fast(image, text)
        out = self.forward_backward(image, text, True, True, _defer_unpack=True, _early_adam=C.addressof(self._ea), **kw)
        if self._ea_ran.value:      # the step updated image_decoder.* / text_decoder.*: the rest, and the step count
            call("mmvae_adam_step_packed_ranges", ptr(st.params), ptr(st.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), st.nparams,
                 self._ea_late, self._ea_nlate, 1, ptr(self.adam_state), self.lr, self.betas[0], self.betas[1], self.eps, 1.0,
                 ptr(st.grad_map()), ptr(st.gpk), ptr(st.gpk_vec), _stream())
        else:
            call("mmvae_adam_step_packed", ptr(st.params), ptr(st.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), st.nparams,
                 ptr(self.adam_state), self.lr, self.betas[0], self.betas[1], self.eps, 1.0, ptr(st.grad_map()), ptr(st.gpk),
                 ptr(st.gpk_vec), _stream())
        self._after_update()
        return out

    def _call_fast(self, image, text) -> StepOutputs:
        """The training loop's call with default arguments: the step-io block and the optimizer calls' argument lists are built
        once and only the fields that change are written (input pointers, pack_first, the stream).  The enqueue thread's Python
        time per step is what makes a loader-fed loop host-bound: 49 launches are 250 us of runtime calls,
        filling a 30-field ctypes structure and converting 18 arguments per call was another 60."""
        st = self.state
        fast = getattr(self, "_fast", None)
        key = (self.enc_dropout, self.gru_dropout, self.kl_lambda, self.seed, self.lr)
        if fast is None or fast[0] != key:
            io = StepIO()
            io.ws, io.ws_bytes = self.ws.data_ptr(), self.ws.numel()
            io.step_counter = self.adam_state.data_ptr()
            io.enc_dropout, io.gru_dropout = int(self.enc_dropout), int(self.gru_dropout)
            io.kl_lambda = self.kl_lambda
            self._pass_config(io, None, None, None, "lambda_xy", "lambda_yx")
            io.seed = self.seed
            io.sums = self.sums.data_ptr()
            io.defer_unpack, io.dp_split = 1, 0
            io.early_adam = C.addressof(self._ea)
            lib = _lib.load()
            late = (ptr(st.params), ptr(st.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), st.nparams, self._ea_late, self._ea_nlate, 1,
                    ptr(self.adam_state), self.lr, self.betas[0], self.betas[1], self.eps, 1.0, ptr(st.grad_map()), ptr(st.gpk), ptr(st.gpk_vec))
            full = (ptr(st.params), ptr(st.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), st.nparams, ptr(self.adam_state), self.lr,
                    self.betas[0], self.betas[1], self.eps, 1.0, ptr(st.grad_map()), ptr(st.gpk), ptr(st.gpk_vec))
            fast = self._fast = (key, io, C.byref(io), lib.mmvae_mm_step, lib.mmvae_adam_step_packed_ranges, lib.mmvae_adam_step_packed, late, full, lib)
        _, io, io_ref, f_step, f_late, f_full, late, full, lib = fast
        assert image.is_contiguous() and text.is_contiguous() and image.dtype == torch.float32 and text.dtype == torch.int64
        assert image.shape[0] == self.B and text.shape == (self.B, 4)
        self._last = ((True, True, True), self._LA, self._LB)
        io.image, io.text = image.data_ptr(), text.data_ptr()
        io.pack_first = int(st.pack_pending)
        stream = _stream()
        rc = f_step(self.h, io_ref, 1, 1, stream)
        if rc != 0:
            raise _lib.MMVAEError("mmvae_mm_step failed (%d): %s" % (rc, lib.mmvae_last_error().decode()))
        st.pack_pending = False
        rc = f_late(*late, stream) if self._ea_ran.value else f_full(*full, stream)
        if rc != 0:
            raise _lib.MMVAEError("optimizer launch failed (%d): %s" % (rc, lib.mmvae_last_error().decode()))
        self._after_update()
        return self._outputs()

    def __call__(self, image, text, **kw) -> StepOutputs:
        if self._dp_active() and getattr(self.all_reduce, "overlap", False):
            return self._call_dp_overlap(image, text, **kw)
        return self._call_packed(image, text, **kw)


class FusedMnistStep(_FusedStepBase):
    """The 3-pass step of mnist/train.py:131-147 (all lambdas 1, KL divided by B*784/3, no annealing)."""

    LAMBDA_XY = (1.0, 1.0, 1.0)
    LAMBDA_YX = (1.0, 1.0, 1.0)

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._LA, self._LB = self.LAMBDA_XY, self.LAMBDA_YX
        self._last = ((True, True, True), self.LAMBDA_XY, self.LAMBDA_YX)

    def _outputs(self) -> StepOutputs:
        passes, lxy, lyx = self._last
        return StepOutputs(self.sums, self.B * 784, self.B, 1.0 / (self.B * (784 / 3)), lxy, lyx, passes)

    def forward_backward(self, image, label, training=True, backward=True, eps=None, recon_image=None, recon_text=None,
                         mu=None, logvar=None, passes=None, lambda_xy=None, lambda_yx=None) -> StepOutputs:
        assert image.is_contiguous() and label.is_contiguous() and image.dtype == torch.float32 and label.dtype == torch.int64
        assert image.numel() == self.B * 784 and label.shape == (self.B,)
        io = _lib.MnistStepIO()
        io.ws, io.ws_bytes = self.ws.data_ptr(), self.ws.numel()
        io.step_counter = self.adam_state.data_ptr()
        io.image, io.label = image.data_ptr(), label.data_ptr()
        for k, t in (("eps", eps), ("recon_image", recon_image), ("recon_text", recon_text), ("mu", mu), ("logvar", logvar)):
            setattr(io, k, None if t is None else t.data_ptr())
        self._pass_config(io, passes, lambda_xy, lambda_yx, "lambda_xy", "lambda_yx")
        io.kl_coef = 1.0 / (self.B * (784 / 3))
        io.seed = self.seed
        io.sums = self.sums.data_ptr()
        call("mmvae_mnist_step", self.h, C.byref(io), int(training), int(backward), _stream())
        return self._outputs()


class FusedCelebaStep(_FusedStepBase):
    """The 3-pass step of celeba/train.py:131-147 (loss_function defaults: lambdas 1, kl_lambda 1e-3)."""

    LAMBDA_X = (1.0, 1.0, 1.0)
    LAMBDA_Y = (1.0, 1.0, 1.0)
    N_ATTRS = 18

    def __init__(self, state: PlanState, batch: int, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 kl_lambda: float = 1e-3, seed: int = 1234, world_size: int = 1, all_reduce=None):
        super().__init__(state, batch, lr, betas, eps, seed, world_size, all_reduce)
        self.kl_lambda = kl_lambda
        self.enc_dropout = True
        self._LA, self._LB = self.LAMBDA_X, self.LAMBDA_Y
        self._last = ((True, True, True), self.LAMBDA_X, self.LAMBDA_Y)

    def _outputs(self) -> StepOutputs:
        passes, lx, ly = self._last
        return StepOutputs(self.sums, self.B * 3 * 64 * 64, self.B * self.N_ATTRS, self.kl_lambda / self.B, lx, ly, passes)

    def __call__(self, image, attrs, **kw) -> StepOutputs:
        return self._call_packed(image, attrs, **kw)

    def forward_backward(self, image, attrs, training=True, backward=True, eps=None, enc_mask=None, recon_image=None,
                         recon_attrs=None, mu=None, logvar=None, passes=None, lambda_x=None, lambda_y=None,
                         _defer_unpack=False) -> StepOutputs:
        assert image.is_contiguous() and attrs.is_contiguous() and image.dtype == torch.float32 and attrs.dtype == torch.float32
        assert image.shape == (self.B, 3, 64, 64) and attrs.shape == (self.B, self.N_ATTRS)
        io = _lib.CelebaStepIO()
        io.ws, io.ws_bytes = self.ws.data_ptr(), self.ws.numel()
        io.step_counter = self.adam_state.data_ptr()
        io.image, io.attrs = image.data_ptr(), attrs.data_ptr()
        for k, t in (("eps", eps), ("enc_mask", enc_mask), ("recon_image", recon_image), ("recon_attrs", recon_attrs),
                     ("mu", mu), ("logvar", logvar)):
            setattr(io, k, None if t is None else t.data_ptr())
        io.enc_dropout = int(self.enc_dropout)
        io.kl_lambda = self.kl_lambda
        self._pass_config(io, passes, lambda_x, lambda_y, "lambda_x", "lambda_y")
        io.seed = self.seed
        io.sums = self.sums.data_ptr()
        io.defer_unpack = int(bool(_defer_unpack))
        call("mmvae_celeba_step", self.h, C.byref(io), int(training), int(backward), _stream())
        return self._outputs()


class FusedCocoStep(_FusedStepBase):
    """The 3-pass step of coco/train.py:138-173 (lambda_xy = (1,1,0), lambda_yx = (1,1,1), kl_lambda 1e-3)."""

    LAMBDA_XY = (1.0, 1.0, 0.0)
    LAMBDA_YX = (1.0, 1.0, 1.0)
    EMB = 300
    _DEFER_PACK = True          # the step refreshes the packed weights itself, the two halves on their own streams

    def __init__(self, state: PlanState, batch: int, sos: torch.Tensor, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 kl_lambda: float = 1e-3, seed: int = 1234, world_size: int = 1, all_reduce=None):
        super().__init__(state, batch, lr, betas, eps, seed, world_size, all_reduce)
        self.kl_lambda = kl_lambda
        self.T = state.steps
        self.sos = sos.to(device=state.device, dtype=torch.float32).contiguous().reshape(self.EMB)
        self.enc_dropout = self.gru_dropout = True
        self._LA, self._LB = self.LAMBDA_XY, self.LAMBDA_YX
        self._last = ((True, True, True), self.LAMBDA_XY, self.LAMBDA_YX)

    def _outputs(self) -> StepOutputs:
        passes, lxy, lyx = self._last
        return StepOutputs(self.sums, self.B * 3 * 32 * 32, self.B * self.T * self.EMB, self.kl_lambda / self.B, lxy, lyx, passes)

    def __call__(self, image, text, **kw) -> StepOutputs:
        return self._call_packed(image, text, **kw)

    def forward_backward(self, image, text, training=True, backward=True, eps=None, enc_mask1=None, enc_mask2=None,
                         gru_keep=None, recon_image=None, recon_text=None, mu=None, logvar=None, passes=None,
                         lambda_xy=None, lambda_yx=None, _defer_unpack=False) -> StepOutputs:
        assert image.is_contiguous() and text.is_contiguous() and image.dtype == torch.float32 and text.dtype == torch.float32
        assert image.shape == (self.B, 3, 32, 32) and text.shape == (self.B, self.T, self.EMB)
        io = _lib.CocoStepIO()
        io.ws, io.ws_bytes = self.ws.data_ptr(), self.ws.numel()
        io.step_counter = self.adam_state.data_ptr()
        io.optimizer_state = self.adam_state.data_ptr() if backward else None     # a void step (exchange time-out) skips its Adam update
        io.image, io.text, io.sos = image.data_ptr(), text.data_ptr(), self.sos.data_ptr()
        for k, t in (("eps", eps), ("enc_mask1", enc_mask1), ("enc_mask2", enc_mask2), ("gru_keep", gru_keep),
                     ("recon_image", recon_image), ("recon_text", recon_text), ("mu", mu), ("logvar", logvar)):
            setattr(io, k, None if t is None else t.data_ptr())
        io.enc_dropout, io.gru_dropout = int(self.enc_dropout), int(self.gru_dropout)
        io.kl_lambda = self.kl_lambda
        self._pass_config(io, passes, lambda_xy, lambda_yx, "lambda_xy", "lambda_yx")
        io.seed = self.seed
        io.defer_unpack = int(bool(_defer_unpack))
        io.pack_first = int(self.state.pack_pending)
        io.sums = self.sums.data_ptr()
        call("mmvae_coco_step", self.h, C.byref(io), int(training), int(backward), _stream())
        self.state.pack_pending = False
        return self._outputs()
