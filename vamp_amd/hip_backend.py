"""Thin, numpy-typed wrapper over the C ABI (one object per ``vamp_ctx``).

This is the only module that touches ctypes pointers; ``vpfits.VPfit`` and the ensemble driver
are written against it.  Every call lands in hand-written HIP (vamp_amd/csrc/vamp_hip.hip).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

MODE_GAUSS3, MODE_VOIGT4, MODE_NBZ3 = 0, 1, 2
Q_OF_MODE = {MODE_GAUSS3: 3, MODE_VOIGT4: 4, MODE_NBZ3: 3}
F64, F32 = 0, 1
WOFZ_ACCURATE, WOFZ_HUMLICEK_W4 = 0, 1


def resolve_dtype(dtype=None):
    """F64 / F32 from None, 0 / 1, or "f64" / "f32".  None means: the environment's ``VAMP_DTYPE`` (SURVEY section 5
    "Config / flags": an override for callers that never pass a type -- the reference's own call sites,
    vpregion.py:59, vpspectrum.py:279, do_vamp.py:59-60), else fp64, the reference's arithmetic width."""
    import os
    if dtype is None:
        dtype = os.environ.get("VAMP_DTYPE") or F64
    if isinstance(dtype, str):
        key = dtype.strip().lower()
        if key in ("f64", "fp64", "float64", "double", "0"):
            return F64
        if key in ("f32", "fp32", "float32", "single", "1"):
            return F32
        raise ValueError("dtype must be 'f64' or 'f32', got %r" % (dtype,))
    if int(dtype) not in (F64, F32):
        raise ValueError("dtype must be F64 (0) or F32 (1), got %r" % (dtype,))
    return int(dtype)


def _dp(a):
    return a.ctypes.data_as(_lib.c_double_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = C.c_int(0)
    _lib.check(_lib.load().vamp_device_count(C.byref(n)))
    return n.value


def comm_unique_id(lib=None):
    """128-byte RCCL id for vamp_comm_init_rank: rank 0 creates it and sends it to the other
    ranks over any host channel."""
    lib = lib if lib is not None else _lib.load()
    buf = C.create_string_buffer(128)
    _lib.check(lib.vamp_comm_unique_id(buf), lib)
    return buf.raw


def comm_library(lib=None):
    """File name of the library whose ncclAllGather carries the in-library exchange (vamp_comm_library)."""
    lib = lib if lib is not None else _lib.load()
    buf = C.create_string_buffer(4096)
    _lib.check(lib.vamp_comm_library(buf, 4096), lib)
    return buf.value.decode("utf-8", "replace")


def default_split_block(W, world=1):
    """Largest even divisor of W that is <= 1024 and keeps W/block a multiple of ``world``."""
    for b in range(min(W, 1024), 1, -1):
        if b % 2 == 0 and W % b == 0 and (W // b) % world == 0:
            return b
    raise ValueError(f"no valid split block for W={W}, world={world}")


class HipContext:
    """One device context: regions + (optionally) an ensemble sampler."""

    def __init__(self, device=0, dtype=None, wofz_kind=None, lib=None):
        """``lib``: an already bound implementation of the C ABI (``_lib.bind(path)``); the default
        -- and the only thing the product uses -- is libvamp_hip.so.  ``dtype``: F64 / F32 / "f64" / "f32";
        None = $VAMP_DTYPE, else fp64."""
        self._lib = lib if lib is not None else _lib.load()
        dtype = resolve_dtype(dtype)
        if wofz_kind is None:
            wofz_kind = WOFZ_ACCURATE if dtype == F64 else WOFZ_HUMLICEK_W4
        h = C.c_void_p()
        self._check(self._lib.vamp_ctx_create(C.byref(h), device, dtype, wofz_kind))
        self._h = h
        self.device = device
        self.dtype = dtype
        self.n_regions = 0
        self.ndims = []
        self.W = 0
        self.comm = None          # (rank, world) once vamp_comm_init_rank has succeeded

    def _check(self, rc):
        _lib.check(rc, self._lib)

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.vamp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, stream_ptr):
        self._check(self._lib.vamp_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_stream_default(self):
        """Run on HIP's legacy default stream (what torch uses when no stream is set)."""
        self._check(self._lib.vamp_ctx_set_stream_default(self._h))

    def synchronize(self):
        self._check(self._lib.vamp_ctx_synchronize(self._h))

    def set_option(self, name, value):
        """Run-time switches (vamp_ctx_set_option): "map_device", "resident", "class_streams"."""
        self._check(self._lib.vamp_ctx_set_option(self._h, name.encode(), int(value)))

    # -- multi-GPU (walker sharding) ---------------------------------------------------------
    def comm_init_rank(self, comm_id, rank, world):
        """Collective: join the RCCL communicator ``comm_id`` (comm_unique_id() of rank 0) as
        ``rank`` of ``world``; the per-half-step exchange then runs inside the library."""
        if len(comm_id) != 128:
            raise ValueError("comm_id must be the 128 bytes of comm_unique_id()")
        self._check(self._lib.vamp_comm_init_rank(self._h, C.c_char_p(bytes(comm_id)), int(rank), int(world)))
        self.comm = (int(rank), int(world))

    def comm_destroy(self):
        self._check(self._lib.vamp_comm_destroy(self._h))
        self.comm = None

    def abandon(self):
        """Drop the handle WITHOUT destroying the context: for a context another thread is still
        inside (a vamp_comm_init_rank that never returned).  The memory is left to process exit."""
        self._h = C.c_void_p()

    def comm_info(self):
        """(rank, world, queried): what the communicator reports through ncclCommUserRank /
        ncclCommCount (queried = True), else the values it was created with."""
        r, w, q = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.vamp_comm_info(self._h, C.byref(r), C.byref(w), C.byref(q)))
        return r.value, w.value, bool(q.value)

    def pack_get(self, part=0):
        """This rank's movers of piece ``part`` after its last half-step: [slots, D + 1]
        (position, lnprob), in slot order (host-staged exchange)."""
        n = self._part_slots
        out = np.empty((n, self.ndims[0] + 1))
        self._check(self._lib.vamp_sampler_pack_get(self._h, int(part), _dp(out)))
        return out

    def scatter_put(self, part, rows_all):
        """Write the movers of piece ``part`` gathered from all ranks (rank order,
        [world * slots, D + 1]) into the state."""
        rows_all = _f64(rows_all)
        if rows_all.shape != (self._shard_world * self._part_slots, self.ndims[0] + 1):
            raise ValueError("rows_all must be [world * part_slots, D + 1]")
        self._check(self._lib.vamp_sampler_scatter_put(self._h, int(part), _dp(rows_all)))

    def set_packing(self, lanes_per_walker):
        """0 = automatic, 16 = four walkers per wavefront (<= 8 components), 64 = one walker per
        wavefront, 65 = one walker per wavefront with its own Taylor tables (<= 8 components),
        256 = one walker per 4-wavefront workgroup (long regions).  Applies from the next
        set_regions call."""
        self._check(self._lib.vamp_ctx_set_packing(self._h, int(lanes_per_walker)))

    # -- data ----------------------------------------------------------------------------
    def set_regions(self, xs, fluxes, noises, n_comp, mode=MODE_VOIGT4, sample_sd=False, include_norm=False,
                    bounds=None, nbz=None):
        """xs/fluxes/noises: lists of 1-D arrays (one per region); n_comp: int or list."""
        if isinstance(xs, np.ndarray) and xs.ndim == 1:
            xs, fluxes, noises = [xs], [fluxes], [noises]
        R = len(xs)
        if np.isscalar(n_comp):
            n_comp = [int(n_comp)] * R
        off = np.zeros(R + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(x) for x in xs])
        x = _f64(np.concatenate([_f64(a) for a in xs]))
        f = _f64(np.concatenate([_f64(a) for a in fluxes]))
        n = _f64(np.concatenate([_f64(a) for a in noises]))
        nc = np.ascontiguousarray(n_comp, dtype=np.int32)
        b = _f64(bounds).reshape(R, 4) if bounds is not None else None
        z = _f64(nbz).reshape(R, 4) if nbz is not None else None
        self._check(self._lib.vamp_set_regions(
            self._h, R, off.ctypes.data_as(_lib.c_int64_p), _dp(x), _dp(f), _dp(n),
            nc.ctypes.data_as(_lib.c_int32_p), int(mode), int(bool(sample_sd)), int(bool(include_norm)), _dp(b), _dp(z)))
        self.n_regions = R
        self.mode = mode
        self.ndims = []
        for r in range(R):
            d = C.c_int(0)
            self._check(self._lib.vamp_region_ndim(self._h, r, C.byref(d)))
            self.ndims.append(d.value)
        self.n_pix = [len(a) for a in xs]
        self.n_comp = [int(k) for k in n_comp]

    def region_classes(self):
        """launch class of every region (vamp_region_class: 0 short, 1 blend, 2 wide, 3 short with <= 2 lines, 4 more
        than 16 lines) and the number of classes of the context"""
        k, n = C.c_int(0), C.c_int(0)
        kinds = []
        for r in range(self.n_regions):
            self._check(self._lib.vamp_region_class(self._h, r, C.byref(k), C.byref(n)))
            kinds.append(k.value)
        return kinds, n.value

    def set_region_ids(self, ids):
        """Global identity of every region in the draw keys (see vamp_set_region_ids)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if ids.size != self.n_regions:
            raise ValueError("one id per region is required")
        self._check(self._lib.vamp_set_region_ids(self._h, ids.ctypes.data_as(_lib.c_int32_p)))
        self.region_ids = [int(i) for i in ids]

    # -- evaluation ----------------------------------------------------------------------
    def lnprob(self, theta, region=0, return_chi2=False):
        theta = _f64(theta)
        if theta.ndim == 1:
            theta = theta[None, :]
        W, D = theta.shape
        if D != self.ndims[region]:
            raise ValueError(f"theta has {D} dims, region {region} needs {self.ndims[region]}")
        out = np.empty(W)
        chi = np.empty(W) if return_chi2 else None
        self._check(self._lib.vamp_lnprob(self._h, region, W, _dp(theta), _dp(out), _dp(chi)))
        return (out, chi) if return_chi2 else out

    def lnprob_all(self, thetas, return_chi2=False):
        """thetas: one [W, D_r] array per region (same W).  Returns lnprob [n_regions, W] (and
        chi2) from a single launch over every region."""
        blocks = [_f64(np.atleast_2d(t)) for t in thetas]
        if len(blocks) != self.n_regions:
            raise ValueError("one theta block per region is required")
        W = blocks[0].shape[0]
        for r, b in enumerate(blocks):
            if b.shape != (W, self.ndims[r]):
                raise ValueError(f"region {r}: theta block must be [{W}, {self.ndims[r]}]")
        flat = np.concatenate([b.ravel() for b in blocks])
        out = np.empty((self.n_regions, W))
        chi = np.empty((self.n_regions, W)) if return_chi2 else None
        self._check(self._lib.vamp_lnprob_all(self._h, W, _dp(flat), _dp(out), _dp(chi)))
        return (out, chi) if return_chi2 else out

    def map_all(self, starts, iterlim=1000, tol=1e-3, active=None, xtol=1e-4, maxfun=0):
        """Nelder-Mead MAP search of every (active) region at once (vamp_map_all).  starts: one
        D_r-vector per region.  ``tol`` is fmin's ftol; ``xtol`` and ``maxfun`` default to scipy's
        1e-4 and 200 evaluations per dimension (maxfun = 0), which is how PyMC 2's
        ``MAP.fit(iterlim, tol)`` calls fmin (vpfits.py:358).  Returns (best [list of D_r-vectors],
        lnprob [n_regions], chi2 [n_regions], iterations [n_regions])."""
        vecs = [_f64(np.ravel(t)) for t in starts]
        if len(vecs) != self.n_regions or any(v.size != d for v, d in zip(vecs, self.ndims)):
            raise ValueError("one start vector of the region's dimension per region is required")
        flat = np.concatenate(vecs)
        best = np.empty_like(flat)
        lnp, chi = np.empty(self.n_regions), np.empty(self.n_regions)
        its = np.zeros(self.n_regions, dtype=np.int64)
        act = None
        if active is not None:
            act = np.ascontiguousarray(np.asarray(active, dtype=np.uint8))
            if act.size != self.n_regions:
                raise ValueError("active needs one flag per region")
        self._check(self._lib.vamp_map_all(self._h, _dp(flat), None if act is None else act.ctypes.data_as(C.c_void_p),
                                          int(iterlim), int(maxfun), float(xtol), float(tol), _dp(best), _dp(lnp),
                                          _dp(chi), its.ctypes.data_as(_lib.c_int64_p)))
        offs = np.concatenate([[0], np.cumsum(self.ndims)])
        return [best[offs[r]:offs[r + 1]].copy() for r in range(self.n_regions)], lnp, chi, its

    def model(self, theta1, region=0):
        theta1 = _f64(theta1).ravel()
        if theta1.size != self.ndims[region]:
            raise ValueError("theta1 has the wrong length")
        K, P = self.n_comp[region], self.n_pix[region]
        tau = np.empty((K, P))
        flux = np.empty(P)
        self._check(self._lib.vamp_model(self._h, region, _dp(theta1), _dp(tau), _dp(flux)))
        return tau, flux

    def model_all(self, thetas):
        """(tau_comp, flux) of every region from ONE launch: lists of [K_r, P_r] and [P_r] arrays for
        one D_r-vector per region."""
        vecs = [_f64(np.ravel(t)) for t in thetas]
        if len(vecs) != self.n_regions or any(v.size != d for v, d in zip(vecs, self.ndims)):
            raise ValueError("one parameter vector of the region's dimension per region is required")
        flat = np.concatenate(vecs)
        sizes = [k * p for k, p in zip(self.n_comp, self.n_pix)]
        tau = np.empty(sum(sizes))
        flux = np.empty(sum(self.n_pix))
        self._check(self._lib.vamp_model_all(self._h, _dp(flat), _dp(tau), _dp(flux)))
        to, po = np.concatenate([[0], np.cumsum(sizes)]), np.concatenate([[0], np.cumsum(self.n_pix)])
        return ([tau[to[r]:to[r + 1]].reshape(self.n_comp[r], self.n_pix[r]) for r in range(self.n_regions)],
                [flux[po[r]:po[r + 1]] for r in range(self.n_regions)])

    def line_records(self, theta1, region=0):
        """(rec[K,5] = centroid, x-scale, y, tau scale, pole factor;  log-prior) as staged on device"""
        theta1 = _f64(theta1).ravel()
        if theta1.size != self.ndims[region]:
            raise ValueError("theta1 has the wrong length")
        rec = np.empty((self.n_comp[region], 5))
        lp = C.c_double(0.0)
        self._check(self._lib.vamp_line_records(self._h, region, _dp(theta1), _dp(rec), C.byref(lp)))
        return rec, lp.value

    def wofz_re(self, x, y):
        x = _f64(x).ravel()
        y = _f64(y).ravel()
        out = np.empty_like(x)
        self._check(self._lib.vamp_wofz_re(self._h, x.size, _dp(x), _dp(y), _dp(out)))
        return out

    # -- sampler -------------------------------------------------------------------------
    def sampler_bind_state(self, X_ptr, lnp_ptr):
        self._check(self._lib.vamp_sampler_bind_state(self._h, C.c_void_p(X_ptr), C.c_void_p(lnp_ptr)))

    def sampler_init(self, theta0, seed=0, a=2.0, split_block=None):
        """theta0: array [W, D] (single region) or list of [W, D_r] arrays."""
        blocks = [theta0] if isinstance(theta0, np.ndarray) else list(theta0)
        if len(blocks) != self.n_regions:
            raise ValueError("one theta0 block per region is required")
        W = blocks[0].shape[0]
        for r, b in enumerate(blocks):
            if b.shape != (W, self.ndims[r]):
                raise ValueError(f"theta0[{r}] must have shape ({W}, {self.ndims[r]})")
        if split_block is None:
            split_block = default_split_block(W)
        flat = _f64(np.concatenate([_f64(b).ravel() for b in blocks]))
        self._check(self._lib.vamp_sampler_init(self._h, W, _dp(flat), C.c_uint64(seed & (2**64 - 1)), float(a), int(split_block)))
        self.W = W
        self.split_block = split_block
        self.total_theta = flat.size
        self.total_walkers = W * self.n_regions

    def sampler_set_shard(self, rank, world):
        return self.sampler_set_shard_parts(rank, world, 1)[0]

    def sampler_set_shard_parts(self, rank, world, parts):
        """Cut this rank's share into ``parts`` pieces (see vamp_sampler_set_shard_parts); returns
        the list of (own_begin, own_end) row ranges, one per piece."""
        b = (C.c_int64 * parts)()
        e = (C.c_int64 * parts)()
        self._check(self._lib.vamp_sampler_set_shard_parts(self._h, rank, world, parts, b, e))
        self._shard_world = world
        self._part_slots = (int(e[0]) - int(b[0])) // 2      # half of every owned split chunk moves per half-step
        return [(int(b[i]), int(e[i])) for i in range(parts)]

    def sampler_state_ptrs(self):
        X, L = C.c_void_p(), C.c_void_p()
        tt, tw = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.vamp_sampler_state_ptrs(self._h, C.byref(X), C.byref(L), C.byref(tt), C.byref(tw)))
        return X.value, L.value, tt.value, tw.value

    def half_step(self, half):
        self._check(self._lib.vamp_sampler_half_step(self._h, int(half)))

    def half_step_part(self, half, part):
        self._check(self._lib.vamp_sampler_half_step_part(self._h, int(half), int(part)))

    def half_step_ext(self, active, partner, zz, logu, region=0):
        a = np.ascontiguousarray(active, dtype=np.int32)
        p = np.ascontiguousarray(partner, dtype=np.int32)
        z = _f64(zz)
        u = _f64(logu)
        self._check(self._lib.vamp_sampler_half_step_ext(
            self._h, region, a.size, a.ctypes.data_as(_lib.c_int32_p), p.ctypes.data_as(_lib.c_int32_p), _dp(z), _dp(u)))

    def _split(self, flat, per_walker):
        """flat [.., total] -> list of per-region arrays"""
        out, o = [], 0
        for r in range(self.n_regions):
            n = self.W * (self.ndims[r] if not per_walker else 1)
            blk = flat[..., o:o + n]
            out.append(blk.reshape(flat.shape[:-1] + ((self.W, self.ndims[r]) if not per_walker else (self.W,))))
            o += n
        return out

    def run_flat(self, n_steps, thin=1, store_chain=True):
        """n_steps with the kept samples as the library lays them out: chain [n_keep, total_theta] (the regions'
        [W, D_r] blocks one after the other), lnprob [n_keep, n_regions * W], n_accept [n_regions * W], seconds.
        (None for chain / lnprob with store_chain=False.)"""
        n_keep = n_steps // thin
        chain = np.empty((n_keep, self.total_theta)) if store_chain else None
        lchain = np.empty((n_keep, self.total_walkers)) if store_chain else None
        nacc = np.empty(self.total_walkers, dtype=np.int64)
        sec = C.c_double(0.0)
        self._check(self._lib.vamp_sampler_run(self._h, n_steps, thin, _dp(chain), _dp(lchain),
                                              nacc.ctypes.data_as(_lib.c_int64_p), C.byref(sec)))
        return chain, lchain, nacc, sec.value

    def run(self, n_steps, thin=1, store_chain=True):
        """Returns dict(chain, lnprob, n_accept, seconds); chain/lnprob are arrays for a single
        region ([n_keep, W, D] / [n_keep, W]) or lists of such arrays."""
        chain, lchain, nacc, seconds = self.run_flat(n_steps, thin=thin, store_chain=store_chain)
        res = {"seconds": seconds, "n_accept": nacc if self.n_regions == 1 else self._split(nacc, True)}
        if store_chain:
            ch, lc = self._split(chain, False), self._split(lchain, True)
            res["chain"] = ch[0] if self.n_regions == 1 else ch
            res["lnprob"] = lc[0] if self.n_regions == 1 else lc
        return res

    def run_dev(self, n_steps, thin=1, chain_ptr=None, lnprob_ptr=None):
        """n_steps with the chain written to caller-owned DEVICE memory (raw pointers, e.g.
        ``tensor.data_ptr()``): chain [n_keep, total_theta], lnprob [n_keep, total_walkers].
        Returns the seconds spent in the sampling loop."""
        sec = C.c_double(0.0)
        self._check(self._lib.vamp_sampler_run_dev(self._h, int(n_steps), int(thin), C.c_void_p(chain_ptr or 0),
                                                  C.c_void_p(lnprob_ptr or 0), C.byref(sec)))
        return sec.value

    def get_state(self):
        th = np.empty(self.total_theta)
        lp = np.empty(self.total_walkers)
        na = np.empty(self.total_walkers, dtype=np.int64)
        st = C.c_int64(0)
        self._check(self._lib.vamp_sampler_get_state(self._h, _dp(th), _dp(lp), na.ctypes.data_as(_lib.c_int64_p), C.byref(st)))
        X = self._split(th, False)
        L = self._split(lp, True)
        if self.n_regions == 1:
            return X[0], L[0], na, st.value
        return X, L, self._split(na, True), st.value

    def set_state(self, theta, lnprob, step):
        blocks = [theta] if isinstance(theta, np.ndarray) else list(theta)
        lps = [lnprob] if isinstance(lnprob, np.ndarray) else list(lnprob)
        th = _f64(np.concatenate([_f64(b).ravel() for b in blocks]))
        lp = _f64(np.concatenate([_f64(b).ravel() for b in lps]))
        self._check(self._lib.vamp_sampler_set_state(self._h, _dp(th), _dp(lp), int(step)))

    def kernel_timing(self, enable=True):
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        self._check(self._lib.vamp_kernel_timing(self._h, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def exchange_timing(self):
        """(ms, count) of the exchanges (all-gather + scatter of one piece) timed while
        kernel_timing was on; resets."""
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        self._check(self._lib.vamp_exchange_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
