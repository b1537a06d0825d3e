"""Device engine: owns the plan, the flat parameter / gradient / Adam buffers and
the packed table in HBM, and drives the C-ABI entry points on a HIP stream.

PyTorch is used for device memory, streams and ``torch.distributed`` only; no
torch operator runs on the compute path.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch

from . import lib as _lib
from .plan import BatchT, ElboOutT, MAX_HIDDEN, ModelLayout, NoiseT


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class PackedTable:
    """The mixed-type table resident in HBM in packed form (include/suave_hip.h)."""

    row_base = 0        # (class defaults: objects assembled field by field -- bench.py, tests -- are whole tables)
    _n_global = None

    @property
    def n_global(self) -> int:
        return self.n_rows if self._n_global is None else self._n_global

    def __init__(self, xnum, codes, mask, device, *, row_base: int = 0, n_global: Optional[int] = None):
        if isinstance(mask, torch.Tensor):  # already on the device (data.pack_frame_device)
            self.xnum = xnum.to(device=device, dtype=torch.float32).contiguous()
            self.codes = codes.to(device=device, dtype=torch.int16).contiguous()
            self.mask = mask.to(device=device, dtype=torch.uint8).contiguous()
        else:
            self.xnum = torch.as_tensor(np.ascontiguousarray(xnum, np.float32)).to(device)
            self.codes = torch.as_tensor(np.ascontiguousarray(codes, np.int16)).to(device)
            self.mask = torch.as_tensor(np.ascontiguousarray(mask, np.uint8)).to(device)
        self.n_rows = int(self.mask.shape[0])
        if self.xnum.shape[0] != self.n_rows or self.codes.shape[0] != self.n_rows:
            raise ValueError("xnum / codes / mask must have the same number of rows")
        self.ld_num = int(self.xnum.shape[1]) if self.xnum.dim() == 2 else 0
        self.ld_disc = int(self.codes.shape[1]) if self.codes.dim() == 2 else 0
        self.ld_mask = int(self.mask.shape[1])
        # Row-sharded tables (data parallelism): this object holds rows [row_base, row_base +
        # n_rows) of a table of n_global rows.  Batches address rows by their GLOBAL index -- the index that keys every
        # in-kernel generator, so a sharded step draws what the one-GPU step draws -- and the array pointers handed to
        # the library are biased by row_base rows, so that global row g resolves to local row g - row_base.
        self.row_base = int(row_base)
        self._n_global = None if n_global is None else int(n_global)

    def local_range(self, start: int, count: int):
        """[begin, end) (global indices) of the rows of global range [start, start + count) held here."""

        b = max(int(start), self.row_base)
        e = min(int(start) + int(count), self.row_base + self.n_rows)
        return (b, max(e, b))

    def local_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """The entries of ``rows`` (global row ids, any integer dtype) held here, order kept."""

        if self.row_base == 0 and self.n_rows == self.n_global:
            return rows
        return rows[(rows >= self.row_base) & (rows < self.row_base + self.n_rows)]

    def batch(self, row_begin: int, n_rows: int, row_index: Optional[torch.Tensor] = None) -> BatchT:
        b = BatchT()
        rb = self.row_base
        b.xnum = self.xnum.data_ptr() - 4 * rb * self.ld_num if self.xnum.numel() else None
        b.codes = self.codes.data_ptr() - 2 * rb * self.ld_disc if self.codes.numel() else None
        b.mask = self.mask.data_ptr() - rb * self.ld_mask
        b.row_index = _ptr(row_index)
        b.row_begin = int(row_begin)
        b.n_rows = int(n_rows)
        b.ld_num, b.ld_disc, b.ld_mask = self.ld_num, self.ld_disc, self.ld_mask
        if row_index is None and n_rows and (row_begin < rb or row_begin + n_rows > rb + self.n_rows):
            raise ValueError("batch exceeds the table (rows are addressed by their global index)")
        return b


class Engine:
    def __init__(self, layout: ModelLayout, device: str | torch.device = "cuda:0"):
        self.layout = layout
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SuaveHipError("the SUAVE HIP engine needs a GPU device (there is no CPU fallback)")
        self.lib = _lib.load()
        torch.cuda.set_device(self.device)
        self._plan = C.c_void_p()
        cols = layout.cols_array()
        _lib.check(self.lib.suave_plan_create(C.byref(layout.model), cols, C.byref(self._plan)), "suave_plan_create")
        n = layout.n_params
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.adam_t = 0
        self._ws: Optional[torch.Tensor] = None
        self._ws_rows = 0
        self.sums = torch.zeros(4, dtype=torch.float64, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "_plan", None) and self._plan.value:
                self.lib.suave_plan_destroy(self._plan)
                self._plan = C.c_void_p()
        except Exception:
            pass

    # -- parameters ------------------------------------------------------------
    def load_state(self, tensors: Mapping[str, np.ndarray]) -> None:
        self.params.copy_(torch.from_numpy(self.layout.pack(tensors)))

    def state(self) -> Dict[str, np.ndarray]:
        return self.layout.unpack(self.params.cpu().numpy())

    def grad_state(self) -> Dict[str, np.ndarray]:
        return self.layout.unpack(self.grads.cpu().numpy())

    def reset_optimizer(self) -> None:
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.adam_t = 0

    # -- workspace -------------------------------------------------------------
    def workspace(self, rows: int) -> torch.Tensor:
        """The caller-owned scratch of the step / decode entry points, grown on demand.  Its size depends on the
        process-wide GEMM arithmetic and small-batch settings (include/suave_hip.h: "size workspaces after changing
        it"), so a change of either -- the automatic switch to the wide arithmetic included -- re-sizes it."""

        mode, small = C.c_int(0), C.c_int(0)
        self.lib.suave_gemm_arithmetic_get(C.byref(mode))
        self.lib.suave_small_batch_rows_get(C.byref(small))
        stamp = (mode.value, small.value)
        if self._ws is None or rows > self._ws_rows or stamp != getattr(self, "_ws_stamp", None):
            rows = max(int(rows), int(self._ws_rows or 0))
            nbytes = C.c_size_t()
            _lib.check(self.lib.suave_workspace_bytes(self._plan, int(rows), C.byref(nbytes)), "suave_workspace_bytes")
            if self._ws is None or int(nbytes.value) + 256 > self._ws.numel():
                self._ws = torch.empty(int(nbytes.value) + 256, dtype=torch.uint8, device=self.device)
            self._ws_rows = rows
            self._ws_stamp = stamp
        return self._ws

    # -- data formats either side of the path (f1 / f2) -----------------------------
    def pack_columns(self, desc: torch.Tensor, n_rows: int, xnum: torch.Tensor, codes: torch.Tensor, mask: torch.Tensor,
                     rejected: torch.Tensor) -> None:
        """suave_pack_columns: ``desc`` is the device image of ``n_cols`` suave_pack_col_t records."""

        _lib.check(self.lib.suave_pack_columns(self._plan, desc.data_ptr(), int(n_rows), _ptr(xnum), int(xnum.stride(0)),
                                               _ptr(codes), int(codes.stride(0)), mask.data_ptr(), int(mask.stride(0)),
                                               rejected.data_ptr(), self._stream()), "suave_pack_columns")

    def unpack_columns(self, desc: torch.Tensor, n_out: int, n_rows: int, out_num: torch.Tensor, out_codes: torch.Tensor,
                       blocks, ld_rows: int) -> None:
        """suave_unpack_columns into the four columnar device blocks (float32, int64, int16, int8)."""

        _lib.check(self.lib.suave_unpack_columns(self._plan, desc.data_ptr(), int(n_out), int(n_rows), _ptr(out_num),
                                                 int(out_num.stride(0)) if out_num.numel() else self.layout.n_num,
                                                 _ptr(out_codes),
                                                 int(out_codes.stride(0)) if out_codes.numel() else self.layout.n_disc,
                                                 blocks[0].data_ptr(), blocks[1].data_ptr(), blocks[2].data_ptr(),
                                                 blocks[3].data_ptr(), int(ld_rows), self._stream()), "suave_unpack_columns")

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _noise(noise: Optional[Mapping[str, object]], seed: int, step: int, keep_alive: list) -> NoiseT:
        nz = NoiseT()
        nz.seed, nz.step = int(seed), int(step)
        if noise:
            for key in ("exp_noise", "eps"):
                t = noise.get(key)
                if t is not None:
                    keep_alive.append(t)
                    setattr(nz, key, t.data_ptr())
            for key in ("enc_keep", "dec_keep"):
                lst = noise.get(key)
                if lst is not None:
                    arr = getattr(nz, key)
                    for i, t in enumerate(lst):
                        keep_alive.append(t)
                        arr[i] = t.data_ptr()
        return nz

    # -- ELBO step -------------------------------------------------------------
    def elbo(self, table: PackedTable, row_begin: int, n_rows: int, *, row_index: Optional[torch.Tensor] = None,
             beta_scale: float, tau: Optional[float] = 1.0, grad: bool = True, training: bool = True,
             global_rows: Optional[int] = None, noise: Optional[Mapping[str, object]] = None, seed: int = 0,
             step: int = 0, want: Sequence[str] = (), accumulate_sums: bool = False,
             dlatent: Optional[torch.Tensor] = None, ready_event: Optional[torch.cuda.Event] = None,
             head=None) -> Dict[str, torch.Tensor]:
        """One ELBO forward (+ backward into ``self.grads``).  ``want`` selects
        per-row outputs among reconstruction, categorical_kl, gaussian_kl,
        latent, component, enc_out, and (supervised branch) the per-column log_px / log_px_missing [B, n_cols].  ``ready_event`` (grad only): recorded by the library once the
        decoder-side gradient bucket is final, before the encoder backward (suave_elbo_backward_overlap).  ``head``
        (grad only; a ``plan.HeadT``): the joint fine-tuning step in one pass -- the classifier head's cross-entropy on
        this step's latent joins the backward (suave_elbo_joint_backward)."""

        B = int(n_rows)
        unsup = self.layout.behaviour != "supervised"
        ws = self.workspace(B * (self.layout.n_components if unsup else 1))
        if tau is None:
            tau = 0.0  # unsupervised branch: temperature None -> plain softmax weights
        keep: list = []
        nz = self._noise(noise, seed, step, keep)
        batch = table.batch(row_begin, B, row_index)
        out = ElboOutT()
        res: Dict[str, torch.Tensor] = {}
        K, D = self.layout.n_components, self.layout.latent_dim
        shapes = {"reconstruction": ((B,), torch.float32), "categorical_kl": ((B,), torch.float32),
                  "gaussian_kl": ((B,), torch.float32), "latent": ((B, D), torch.float32),
                  "component": ((B,), torch.int32), "enc_out": ((B, K + 2 * K * D), torch.float32),
                  "log_px": ((B, self.layout.model.n_cols), torch.float32),
                  "log_px_missing": ((B, self.layout.model.n_cols), torch.float32)}
        for name in want:
            shape, dt = shapes[name]
            res[name] = torch.empty(shape, dtype=dt, device=self.device)
            setattr(out, name, res[name].data_ptr())
        if accumulate_sums:
            out.sums = self.sums.data_ptr()
        base = (ws.data_ptr() + 255) // 256 * 256
        nbytes = ws.numel() - (base - ws.data_ptr())
        if grad and head is not None:
            rc = self.lib.suave_elbo_joint_backward(self._plan, self.params.data_ptr(), C.byref(batch), C.byref(nz),
                                                    float(beta_scale), float(tau), int(global_rows or B), base, nbytes,
                                                    self._stream(), C.byref(out), self.grads.data_ptr(), C.byref(head),
                                                    int(ready_event.cuda_event) if ready_event is not None else None)
            _lib.check(rc, "suave_elbo_joint_backward")
        elif grad and ready_event is not None:
            rc = self.lib.suave_elbo_backward_overlap(self._plan, self.params.data_ptr(), C.byref(batch), C.byref(nz),
                                                      float(beta_scale), float(tau), int(global_rows or B), base, nbytes,
                                                      self._stream(), C.byref(out), self.grads.data_ptr(), _ptr(dlatent),
                                                      int(ready_event.cuda_event))
            _lib.check(rc, "suave_elbo_backward_overlap")
        elif grad:
            rc = self.lib.suave_elbo_backward(self._plan, self.params.data_ptr(), C.byref(batch), C.byref(nz),
                                              float(beta_scale), float(tau), int(global_rows or B), base, nbytes,
                                              self._stream(), C.byref(out), self.grads.data_ptr(), _ptr(dlatent))
            _lib.check(rc, "suave_elbo_backward")
        else:
            rc = self.lib.suave_elbo_forward(self._plan, self.params.data_ptr(), C.byref(batch), C.byref(nz),
                                             float(beta_scale), float(tau), int(bool(training)), base, nbytes,
                                             self._stream(), C.byref(out))
            _lib.check(rc, "suave_elbo_forward")
        return res

    def adam_step(self, lr: float | Mapping[str, float], *, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        """torch.optim.Adam step; ``lr`` may map group name (encoder/decoder/prior) -> lr (0 = frozen)."""

        self.adam_t += 1
        names = ["encoder", "decoder", "prior"]
        begins = [self.layout.group_ranges[n][0] for n in names] + [self.layout.group_ranges["prior"][1]]
        lrs = [float(lr[n]) if isinstance(lr, Mapping) else float(lr) for n in names]
        gb = (C.c_int32 * 4)(*begins)
        lra = (C.c_float * 3)(*lrs)
        rc = self.lib.suave_adam_step(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), gb, lra, 3, self.adam_t, float(betas[0]),
                                      float(betas[1]), float(eps), self._stream())
        _lib.check(rc, "suave_adam_step")

    def head_step(self, head, latent: torch.Tensor, rows: torch.Tensor, *, logits: Optional[torch.Tensor] = None) -> None:
        """suave_head_step: one batch of the classifier-head phase on cached latents -- ``rows`` (int32) are row ids into
        ``latent`` [N, D] and into the head's targets; ``head`` is a ``plan.HeadT``."""

        rc = self.lib.suave_head_step(C.byref(head), int(latent.shape[1]), latent.data_ptr(), int(latent.stride(0)),
                                      rows.data_ptr(), 0, int(rows.numel()), self._stream(), _ptr(logits))
        _lib.check(rc, "suave_head_step")

    def adam_flat(self, params: torch.Tensor, grads: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step_t: int, lr: float,
                  *, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        """torch.optim.Adam step on one flat buffer with one learning rate (the classifier head's group)."""

        gb = (C.c_int32 * 2)(0, int(params.numel()))
        lra = (C.c_float * 1)(float(lr))
        rc = self.lib.suave_adam_step(params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), gb, lra, 1, int(step_t),
                                      float(betas[0]), float(betas[1]), float(eps), self._stream())
        _lib.check(rc, "suave_adam_step")

    # -- encode ----------------------------------------------------------------
    def encode(self, table: PackedTable, row_begin: int, n_rows: int, *, temperature: Optional[float] = None,
               row_index: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        B = int(n_rows)
        K, D = self.layout.n_components, self.layout.latent_dim
        ws = self.workspace(B)
        base = (ws.data_ptr() + 255) // 256 * 256
        nbytes = ws.numel() - (base - ws.data_ptr())
        batch = table.batch(row_begin, B, row_index)
        enc = torch.empty((B, K + 2 * K * D), dtype=torch.float32, device=self.device)
        pm = torch.empty((B, D), dtype=torch.float32, device=self.device)
        plv = torch.empty((B, D), dtype=torch.float32, device=self.device)
        pr = torch.empty((B, K), dtype=torch.float32, device=self.device)
        inv_t = 1.0 if temperature is None else 1.0 / max(float(temperature), 1e-6)
        rc = self.lib.suave_encode(self._plan, self.params.data_ptr(), C.byref(batch), inv_t, base, nbytes,
                                   self._stream(), enc.data_ptr(), pm.data_ptr(), plv.data_ptr(), pr.data_ptr())
        _lib.check(rc, "suave_encode")
        return {"enc_out": enc, "mean": pm, "logvar": plv, "probs": pr}

    def encode_hard(self, table: PackedTable, row_begin: int, n_rows: int, *, temperature: Optional[float] = None):
        """The hard assignment of SUAVE.impute (model.py:3627-3634): (z = mu of the most probable component [B,D],
        component [B]) straight from the library (suave_encode_hard)."""

        B = int(n_rows)
        ws = self.workspace(B)
        base = (ws.data_ptr() + 255) // 256 * 256
        nbytes = ws.numel() - (base - ws.data_ptr())
        batch = table.batch(row_begin, B, None)
        z = torch.empty((B, self.layout.latent_dim), dtype=torch.float32, device=self.device)
        comp = torch.empty((B,), dtype=torch.int32, device=self.device)
        inv_t = 1.0 if temperature is None else 1.0 / max(float(temperature), 1e-6)
        rc = self.lib.suave_encode_hard(self._plan, self.params.data_ptr(), C.byref(batch), inv_t, base, nbytes,
                                        self._stream(), comp.data_ptr(), z.data_ptr())
        _lib.check(rc, "suave_encode_hard")
        return z, comp

    # -- generation ------------------------------------------------------------
    def prior_latents(self, n: int, cdf: torch.Tensor, *, u=None, eps=None, seed: int = 0, row_offset: int = 0):
        D = self.layout.latent_dim
        z = torch.empty((n, D), dtype=torch.float32, device=self.device)
        comp = torch.empty((n,), dtype=torch.int32, device=self.device)
        rc = self.lib.suave_prior_latents(self._plan, self.params.data_ptr(), cdf.data_ptr(), _ptr(u), _ptr(eps),
                                          int(seed), int(row_offset), int(n), self._stream(), z.data_ptr(),
                                          comp.data_ptr())
        _lib.check(rc, "suave_prior_latents")
        return z, comp

    def conditional_latents(self, labels: torch.Tensor, class_offsets: torch.Tensor, class_rows: torch.Tensor,
                            train_probs: torch.Tensor, train_mu: torch.Tensor, train_logvar: torch.Tensor, *,
                            u_row=None, e_comp=None, eps=None, seed: int = 0, row_offset: int = 0):
        n = int(labels.numel())
        D = self.layout.latent_dim
        z = torch.empty((n, D), dtype=torch.float32, device=self.device)
        comp = torch.empty((n,), dtype=torch.int32, device=self.device)
        rc = self.lib.suave_conditional_latents(self._plan, class_offsets.data_ptr(), class_rows.data_ptr(),
                                                labels.data_ptr(), train_probs.data_ptr(), train_mu.data_ptr(),
                                                train_logvar.data_ptr(), _ptr(u_row), _ptr(e_comp), _ptr(eps),
                                                int(seed), int(row_offset), n, self._stream(), z.data_ptr(),
                                                comp.data_ptr())
        _lib.check(rc, "suave_conditional_latents")
        return z, comp

    def decode_sample(self, z: torch.Tensor, *, component: Optional[torch.Tensor] = None,
                      assignments: Optional[torch.Tensor] = None, noise_normal=None, noise_uniform=None,
                      noise_exp=None, seed: int = 0, row_offset: int = 0, want_y: bool = False):
        n = int(z.shape[0])
        L = self.layout
        ws = self.workspace(n)
        base = (ws.data_ptr() + 255) // 256 * 256
        nbytes = ws.numel() - (base - ws.data_ptr())
        out_num = torch.empty((n, max(L.n_num, 1)), dtype=torch.float32, device=self.device)
        out_codes = torch.empty((n, max(L.n_disc, 1)), dtype=torch.int16, device=self.device)
        y_out = torch.empty((n, L.y_rows), dtype=torch.float32, device=self.device) if want_y else None
        rc = self.lib.suave_decode_sample(self._plan, self.params.data_ptr(), z.data_ptr(), _ptr(assignments),
                                          _ptr(component), n, _ptr(noise_normal), _ptr(noise_uniform),
                                          _ptr(noise_exp), int(seed), int(row_offset), base, nbytes, self._stream(),
                                          out_num.data_ptr(), int(out_num.shape[1]), out_codes.data_ptr(),
                                          int(out_codes.shape[1]), _ptr(y_out))
        _lib.check(rc, "suave_decode_sample")
        if want_y:
            if L.y_rows != L.total_y:  # internal layout (wide categorical slots) -> the reference's y_full columns
                y_out = y_out[:, torch.as_tensor(L.y_index, device=self.device)]
            return out_num[:, : L.n_num], out_codes[:, : L.n_disc], y_out
        return out_num[:, : L.n_num], out_codes[:, : L.n_disc]

    def decode_sample_static(self, z: torch.Tensor, out_num: torch.Tensor, out_codes: torch.Tensor, *,
                             component: Optional[torch.Tensor] = None, assignments: Optional[torch.Tensor] = None,
                             noise_normal=None, noise_uniform=None, noise_exp=None, seed: int = 0, row_offset: int = 0,
                             y_out: Optional[torch.Tensor] = None) -> None:
        """suave_decode_sample_static: the decode for callers that cannot allocate.  Every buffer, outputs
        included, comes from the caller; the library's scratch is its own static device array, so this method
        itself creates no tensor (rows are limited by SUAVE_STATIC_SCRATCH_BYTES, include/suave_hip.h)."""

        rc = self.lib.suave_decode_sample_static(self._plan, self.params.data_ptr(), z.data_ptr(), _ptr(assignments),
                                                 _ptr(component), int(z.shape[0]), _ptr(noise_normal),
                                                 _ptr(noise_uniform), _ptr(noise_exp), int(seed), int(row_offset),
                                                 self._stream(), out_num.data_ptr(), int(out_num.shape[1]),
                                                 out_codes.data_ptr(), int(out_codes.shape[1]), _ptr(y_out))
        _lib.check(rc, "suave_decode_sample_static")

    # -- Monte-Carlo posterior predictive ------------------------------------------
    def posterior_draws(self, enc_out: torch.Tensor, L: int, *, e=None, eps=None, seed: int = 0, row_offset: int = 0,
                        row_major: bool = False):
        """SUAVE._posterior_draws: returns z [L*B, D] and component [L*B], draw-major like the reference (row l B + b)
        or, ``row_major``, with each row's draws together (row b L + l).  ``row_offset``: global index of the first
        table row (the generators are keyed by (row, draw))."""

        B = int(enc_out.shape[0])
        D = self.layout.latent_dim
        z = torch.empty((L * B, D), dtype=torch.float32, device=self.device)
        comp = torch.empty((L * B,), dtype=torch.int32, device=self.device)
        fn = self.lib.suave_posterior_draws_by_row if row_major else self.lib.suave_posterior_draws
        rc = fn(self._plan, enc_out.data_ptr(), B, int(L), _ptr(e), _ptr(eps), int(seed), int(row_offset), self._stream(),
                z.data_ptr(), comp.data_ptr())
        _lib.check(rc, "suave_posterior_draws")
        return z, comp

    def decode_log_px(self, z: torch.Tensor, component: torch.Tensor, table: PackedTable, row_begin: int = 0, *,
                      row_index: Optional[torch.Tensor] = None):
        """Decoder.forward's per-column ``log_px`` / ``log_px_missing`` (decoder.py:348-379) of the table rows at
        the given latents and components: two [n, n_cols] tensors in schema column order."""

        n = int(z.shape[0])
        ws = self.workspace(n)
        base = (ws.data_ptr() + 255) // 256 * 256
        nbytes = ws.numel() - (base - ws.data_ptr())
        batch = table.batch(row_begin, n, row_index)
        n_cols = int(self.layout.model.n_cols)
        lp = torch.empty((n, n_cols), dtype=torch.float32, device=self.device)
        lpm = torch.empty((n, n_cols), dtype=torch.float32, device=self.device)
        rc = self.lib.suave_decode_log_px(self._plan, self.params.data_ptr(), z.data_ptr(), component.data_ptr(),
                                          C.byref(batch), base, nbytes, self._stream(), lp.data_ptr(), lpm.data_ptr())
        _lib.check(rc, "suave_decode_log_px")
        return lp, lpm

    def head_params_width(self, col_index: int) -> int:
        w = C.c_int32()
        _lib.check(self.lib.suave_head_params_width(self._plan, int(col_index), C.byref(w)), "suave_head_params_width")
        return int(w.value)

    def decode_params(self, z: torch.Tensor, col_index: int, *, component: Optional[torch.Tensor] = None,
                      assignments: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = int(z.shape[0])
        ws = self.workspace(n)
        base = (ws.data_ptr() + 255) // 256 * 256
        nbytes = ws.numel() - (base - ws.data_ptr())
        out = torch.empty((n, self.head_params_width(col_index)), dtype=torch.float32, device=self.device)
        rc = self.lib.suave_decode_params(self._plan, self.params.data_ptr(), z.data_ptr(), _ptr(assignments),
                                          _ptr(component), n, int(col_index), base, nbytes, self._stream(),
                                          out.data_ptr())
        _lib.check(rc, "suave_decode_params")
        return out
