"""ctypes binding of libmpsfr.so (include/mpsfr.h).  No CPU fallback: if the HIP library is
missing or cannot be loaded this module raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MPSFR_LIB_PATH: load another build of the library (kernel experiments, scripts/variants.py)
LIB_PATH = os.environ.get('MPSFR_LIB_PATH') or os.path.join(HERE, 'libmpsfr.so')

NFIT = 16
FIT_ILL_CONDITIONED = 4      # status bit of fit_out[14] (MPSFR_FIT_ILL_CONDITIONED, include/mpsfr.h)
DIM_AO = 80
PREC_MIXED, PREC_F64 = 0, 1
E_GRID = -3

_lib = None


class MpsfrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libmpsfr error %d: %s' % (code, msg))
        self.code = code


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as
    /opt/rocm's).  If libmpsfr.so pulled in the system runtime first and torch were imported
    later, the process would hold two runtimes and the second finds no GPU.  So when torch is
    installed, its runtime is loaded first (without importing torch) and libmpsfr binds to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load libmpsfr.so (built by muse_psfr_amd._build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s not found: build it with `python -m muse_psfr_amd._build` '
                          '(hipcc, gfx950); there is no CPU fallback' % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    p = C.c_void_p
    dp = C.POINTER(C.c_double)
    u8p = C.POINTER(C.c_uint8)
    lib.mpsfr_create.argtypes = [C.POINTER(p), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
    lib.mpsfr_create.restype = C.c_int
    lib.mpsfr_destroy.argtypes = [p]
    lib.mpsfr_destroy.restype = None
    lib.mpsfr_last_error.argtypes = []
    lib.mpsfr_last_error.restype = C.c_char_p
    lib.mpsfr_set_option.argtypes = [p, C.c_char_p, C.c_double]
    lib.mpsfr_set_option.restype = C.c_int
    lib.mpsfr_reconstruct.argtypes = [p, C.c_int, dp, dp, dp, u8p, dp, C.c_double, C.c_int,
                                      C.c_int, dp, u8p, u8p, p, p, p, C.c_int]
    lib.mpsfr_reconstruct.restype = C.c_int
    lib.mpsfr_reconstruct_multi.argtypes = [C.POINTER(p), C.c_int, C.c_int, dp, dp, dp, u8p, dp, C.c_double,
                                            C.c_int, C.c_int, dp, u8p, u8p, p, p, p]
    lib.mpsfr_reconstruct_multi.restype = C.c_int
    lib.mpsfr_reconstruct_multi_async.argtypes = lib.mpsfr_reconstruct_multi.argtypes
    lib.mpsfr_reconstruct_multi_async.restype = C.c_int
    lib.mpsfr_wait_multi.argtypes = [C.POINTER(p), C.c_int]
    lib.mpsfr_wait_multi.restype = C.c_int
    lib.mpsfr_fit_stamps.argtypes = [p, C.c_int, p, p, C.c_int]
    lib.mpsfr_fit_stamps.restype = C.c_int
    lib.mpsfr_simul_psd.argtypes = [p, C.c_double, C.c_double, C.c_double, C.c_int, dp, C.c_double, C.c_int, u8p, u8p, dp]
    lib.mpsfr_simul_psd.restype = C.c_int
    lib.mpsfr_psf_from_psd.argtypes = [p, C.c_int, dp, C.c_int, dp, dp]
    lib.mpsfr_psf_from_psd.restype = C.c_int
    lib.mpsfr_convolve_stamps.argtypes = [p, C.c_int, dp, dp, dp, C.c_int, dp, dp, dp]
    lib.mpsfr_convolve_stamps.restype = C.c_int
    lib.mpsfr_fit_rows.argtypes = [dp, C.c_long, C.c_double, dp, C.c_long]
    lib.mpsfr_fit_rows.restype = C.c_int
    lib.mpsfr_sync.argtypes = [p]
    lib.mpsfr_sync.restype = C.c_int
    lib.mpsfr_last_ticket.argtypes = [p]
    lib.mpsfr_last_ticket.restype = C.c_long
    lib.mpsfr_wait.argtypes = [p, C.c_long]
    lib.mpsfr_wait.restype = C.c_int
    lib.mpsfr_abandon.argtypes = [p]
    lib.mpsfr_abandon.restype = C.c_int
    lib.mpsfr_stream.argtypes = [p]
    lib.mpsfr_stream.restype = C.c_void_p
    lib.mpsfr_stream_wait.argtypes = [p, C.c_void_p]
    lib.mpsfr_stream_wait.restype = C.c_int
    lib.mpsfr_wait_event.argtypes = [p, C.c_void_p]
    lib.mpsfr_wait_event.restype = C.c_int
    lib.mpsfr_host_time.argtypes = [p, dp, C.POINTER(C.c_long)]
    lib.mpsfr_host_time.restype = C.c_int
    lib.mpsfr_debug_fetch.argtypes = [p, C.c_char_p, dp, C.c_size_t]
    lib.mpsfr_debug_fetch.restype = C.c_long
    lib.mpsfr_profile_count.argtypes = []
    lib.mpsfr_profile_count.restype = C.c_int
    lib.mpsfr_profile_name.argtypes = [C.c_int]
    lib.mpsfr_profile_name.restype = C.c_char_p
    lib.mpsfr_profile_get.argtypes = [p, C.c_int, dp, C.POINTER(C.c_long)]
    lib.mpsfr_profile_get.restype = C.c_int
    lib.mpsfr_profile_reset.argtypes = [p]
    lib.mpsfr_profile_reset.restype = C.c_int
    lib.mpsfr_version.argtypes = []
    lib.mpsfr_version.restype = C.c_int
    lib.mpsfr_device_count.argtypes = []
    lib.mpsfr_device_count.restype = C.c_int
    lib.mpsfr_build_id.argtypes = []
    lib.mpsfr_build_id.restype = C.c_char_p
    _lib = lib
    return lib


EXPORTS = ['mpsfr_create', 'mpsfr_destroy', 'mpsfr_last_error', 'mpsfr_set_option',
           'mpsfr_reconstruct', 'mpsfr_reconstruct_multi', 'mpsfr_reconstruct_multi_async', 'mpsfr_wait_multi', 'mpsfr_fit_stamps', 'mpsfr_simul_psd', 'mpsfr_psf_from_psd',
           'mpsfr_convolve_stamps', 'mpsfr_fit_rows', 'mpsfr_sync', 'mpsfr_last_ticket', 'mpsfr_wait', 'mpsfr_abandon',
           'mpsfr_stream', 'mpsfr_stream_wait', 'mpsfr_wait_event',
           'mpsfr_host_time', 'mpsfr_debug_fetch',
           'mpsfr_profile_count', 'mpsfr_profile_name', 'mpsfr_profile_get',
           'mpsfr_profile_reset', 'mpsfr_version', 'mpsfr_build_id', 'mpsfr_device_count']


def fit_rows(fit, pixscale, out):
    """mpsfr_fit_rows: the 14 fit columns of FIT_ROWS from library fit rows `fit` (n, NFIT) into `out`, a C-contiguous
    float64 (n, stride >= 14) block whose first 14 columns receive them."""
    f = np.ascontiguousarray(fit, dtype=np.float64).reshape(-1, NFIT)
    assert out.dtype == np.float64 and out.ndim == 2 and out.shape[0] == f.shape[0] and out.strides[1] == 8
    _check(load().mpsfr_fit_rows(_dptr(f), f.shape[0], float(pixscale), out.ctypes.data_as(C.POINTER(C.c_double)),
                                 out.strides[0] // 8))


def device_count():
    """HIP devices visible to this process (0 without a GPU)."""
    return int(load().mpsfr_device_count())


def _check(rc):
    if rc < 0:
        raise MpsfrError(rc, load().mpsfr_last_error().decode())
    return rc


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8ptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))


class Context:
    """One GPU context: grid size, precision, HIP stream, workspaces."""

    def __init__(self, dim=1280, pixscale=0.2, dimpsf=40, precision='mixed', device=0):
        self.lib = load()
        self.dim, self.pixscale, self.dimpsf = int(dim), float(pixscale), int(dimpsf)
        self.precision = precision
        prec = {'mixed': PREC_MIXED, 'f64': PREC_F64}[precision]
        h = C.c_void_p()
        _check(self.lib.mpsfr_create(C.byref(h), int(device), self.dim, self.dimpsf,
                                     self.pixscale, prec))
        self._h = h
        # Output arrays of asynchronous host-output calls not yet handed over, by ticket: the library holds
        # raw pointers to them (it writes them in mpsfr_wait, in mpsfr_sync, or when the ring of four comes
        # round again), so the context keeps them alive until then -- a PendingResult that is dropped
        # without wait() leaves no dangling pointer behind.
        self._pending = {}
        self._abandoned = set()

    def close(self):
        if getattr(self, '_h', None):
            self.lib.mpsfr_destroy(self._h)        # (destroys the ring without writing to any caller array)
            self._h = None
            self._pending = {}

    __del__ = close

    def set_option(self, key, value):
        _check(self.lib.mpsfr_set_option(self._h, key.encode(), float(value)))

    def sync(self):
        _check(self.lib.mpsfr_sync(self._h))
        self._pending.clear()

    def abandon(self):
        """Drop every asynchronous host-output call not yet waited for (mpsfr_abandon): the GPU drains, the
        library forgets the output arrays without writing to them, the context stays usable.  For error paths."""
        try:
            _check(self.lib.mpsfr_abandon(self._h))
        finally:
            self._abandoned.update(t for t in self._pending if not isinstance(t, tuple))
            self._pending.clear()

    def _handed_over(self, upto):
        for t in [t for t in self._pending if (t[1] if isinstance(t, tuple) else t) <= upto]:
            del self._pending[t]

    def wait_event(self, hip_event):
        """The next reconstruct call waits on the GPU for this recorded hipEvent_t (an integer
        handle, e.g. torch.cuda.Event.cuda_event)."""
        _check(self.lib.mpsfr_wait_event(self._h, C.c_void_p(int(hip_event))))

    def stream_wait(self, hip_stream):
        """Make the caller's stream (an integer hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) wait on the
        GPU for every call made so far (mpsfr_stream_wait): the cheap form of waiting on stream_handle()."""
        _check(self.lib.mpsfr_stream_wait(self._h, C.c_void_p(int(hip_stream))))

    def host_time(self):
        """(seconds spent inside mpsfr_reconstruct, calls) since the last profile_reset()."""
        sec, n = C.c_double(), C.c_long()
        _check(self.lib.mpsfr_host_time(self._h, C.byref(sec), C.byref(n)))
        return sec.value, n.value

    def stream_handle(self):
        """hipStream_t of the context as an integer (for torch.cuda.ExternalStream)."""
        return int(self.lib.mpsfr_stream(self._h) or 0)

    def reconstruct_async(self, *args, **kwargs):
        """`reconstruct` without waiting for the GPU (on_device = 2): returns a PendingResult whose
        .wait() blocks until the call has finished and returns the same dict.  Up to four such calls
        are in flight per context, overlapping on the pipeline lanes; they complete in order."""
        return self.reconstruct(*args, _async=True, **kwargs)

    def reconstruct(self, lbda, seeing, gl, l0, three_lgs=None, h=(100, 10000), wind_speed=None,
                    npsflin=1, masks=None, want_psf=True, want_sum=True, want_fit=True, _async=False):
        """Host-buffer call.  Returns dict(psf, psf_sum, fit) of float64 arrays (or None)."""
        seeing = np.ascontiguousarray(np.atleast_1d(seeing), dtype=np.float64)
        gl = np.ascontiguousarray(np.atleast_1d(gl), dtype=np.float64)
        l0 = np.ascontiguousarray(np.atleast_1d(l0), dtype=np.float64)
        lbda = np.ascontiguousarray(np.atleast_1d(lbda), dtype=np.float64)
        nt, nl = seeing.size, lbda.size
        three = np.zeros(nt, np.uint8) if three_lgs is None else \
            np.ascontiguousarray(np.atleast_1d(three_lgs)).astype(np.uint8)
        if wind_speed is None:
            wind_speed = float(np.full_like(np.array(h), 12.5)[0])      # psfrec.py:61
        hh = np.ascontiguousarray(h, dtype=np.float64)
        if hh.size != 2:
            raise ValueError('exactly two layers are supported (psfrec.py:66)')
        mrec = mres = None
        if masks is not None:
            mrec = np.ascontiguousarray(masks[0]).astype(np.uint8).reshape(-1)
            mres = np.ascontiguousarray(masks[1]).astype(np.uint8).reshape(-1)
            assert mrec.size == DIM_AO * DIM_AO and mres.size == DIM_AO * DIM_AO
        n = self.dimpsf
        psf = np.empty((nt, nl, n, n)) if want_psf else None
        psum = np.empty((nl, n, n)) if want_sum else None
        fit = np.empty((nt, nl, NFIT)) if want_fit else None
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _check(self.lib.mpsfr_reconstruct(
            self._h, nt, _dptr(seeing), _dptr(gl), _dptr(l0), _u8ptr(three), _dptr(hh),
            float(wind_speed), int(npsflin), nl, _dptr(lbda), _u8ptr(mrec), _u8ptr(mres),
            vp(psf), vp(psum), vp(fit), 2 if _async else 0))
        if _async:
            ticket = int(self.lib.mpsfr_last_ticket(self._h))
            arrays = dict(psf=psf, psf_sum=psum, fit=fit)
            self._pending[ticket] = arrays
            self._handed_over(ticket - 4)        # the call itself handed over the ticket four calls back
            return PendingResult(self, ticket, arrays)
        return dict(psf=psf, psf_sum=psum, fit=fit)

    @staticmethod
    def reconstruct_multi_async(ctxs, *args, **kwargs):
        """`reconstruct_multi` without waiting for the GPUs (mpsfr_reconstruct_multi_async): every context takes its
        shard as an asynchronous host-output call; returns a PendingMulti whose .wait() (mpsfr_wait_multi) returns the
        same dict.  One such call may be pending per ctxs[0]."""
        return Context.reconstruct_multi(ctxs, *args, _async=True, **kwargs)

    @staticmethod
    def reconstruct_multi(ctxs, lbda, seeing, gl, l0, three_lgs=None, h=(100, 10000), wind_speed=None,
                          npsflin=1, masks=None, want_psf=True, want_sum=True, want_fit=True, _async=False):
        """`reconstruct` with the rows in contiguous shards over several contexts (one per device,
        one host thread each inside the library: mpsfr_reconstruct_multi) -- the reference's joblib
        fan-out (psfrec.py:1082-1083).  Same outputs as the single-context call."""
        ctxs = list(ctxs)
        seeing = np.ascontiguousarray(np.atleast_1d(seeing), dtype=np.float64)
        gl = np.ascontiguousarray(np.atleast_1d(gl), dtype=np.float64)
        l0 = np.ascontiguousarray(np.atleast_1d(l0), dtype=np.float64)
        lbda = np.ascontiguousarray(np.atleast_1d(lbda), dtype=np.float64)
        nt, nl = seeing.size, lbda.size
        three = np.zeros(nt, np.uint8) if three_lgs is None else \
            np.ascontiguousarray(np.atleast_1d(three_lgs)).astype(np.uint8)
        if wind_speed is None:
            wind_speed = float(np.full_like(np.array(h), 12.5)[0])      # psfrec.py:61
        hh = np.ascontiguousarray(h, dtype=np.float64)
        if hh.size != 2:
            raise ValueError('exactly two layers are supported (psfrec.py:66)')
        mrec = mres = None
        if masks is not None:
            mrec = np.ascontiguousarray(masks[0]).astype(np.uint8).reshape(-1)
            mres = np.ascontiguousarray(masks[1]).astype(np.uint8).reshape(-1)
            assert mrec.size == DIM_AO * DIM_AO and mres.size == DIM_AO * DIM_AO
        n = ctxs[0].dimpsf
        psf = np.empty((nt, nl, n, n)) if want_psf else None
        psum = np.empty((nl, n, n)) if want_sum else None
        fit = np.empty((nt, nl, NFIT)) if want_fit else None
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        handles = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        fn = ctxs[0].lib.mpsfr_reconstruct_multi_async if _async else ctxs[0].lib.mpsfr_reconstruct_multi
        arrays = dict(psf=psf, psf_sum=psum, fit=fit)
        if _async:
            # the refusals the library makes BEFORE it queues anything are made here, with nothing touched:
            # whatever the library itself reports below is a failed shard, after which it has abandoned every
            # pending asynchronous call of every context of the call
            if any(isinstance(t, tuple) for t in ctxs[0]._pending):
                raise MpsfrError(-1, 'a multi-context call is pending on this context: wait for it first')
            c0 = ctxs[0]
            if any((c.dim, c.dimpsf, c.pixscale, c.precision) != (c0.dim, c0.dimpsf, c0.pixscale, c0.precision)
                   for c in ctxs[1:]):
                raise MpsfrError(-1, 'the contexts must share dim, dimpsf, pixscale and precision')
        try:
            _check(fn(handles, len(ctxs), nt, _dptr(seeing), _dptr(gl), _dptr(l0), _u8ptr(three), _dptr(hh),
                      float(wind_speed), int(npsflin), nl, _dptr(lbda), _u8ptr(mrec), _u8ptr(mres),
                      vp(psf), vp(psum), vp(fit)))
        except MpsfrError:
            if _async:
                # The library abandoned the shards it had queued on ctxs[0..k-1]; earlier reconstruct_async tickets
                # of the failing and the later contexts are still pending in C with pointers into arrays this side
                # would otherwise forget.  abandon() on EVERY context makes the two sides agree: the library drains
                # and drops what it holds, and the tickets land in `_abandoned` so that a later wait() raises
                # instead of handing back arrays nobody wrote.
                for c in ctxs:
                    c.abandon()
            raise
        if _async:
            # the library holds pointers into `arrays` until mpsfr_wait_multi: every context keeps them alive
            # under the ticket of its shard
            for c in ctxs:
                t = int(c.lib.mpsfr_last_ticket(c._h))
                if t >= 0:
                    c._pending[('multi', t)] = arrays
            return PendingMulti(ctxs, handles, arrays)
        return arrays

    def reconstruct_device(self, lbda, seeing, gl, l0, three_lgs, h, wind_speed, npsflin, masks,
                           psf_ptr, sum_ptr, fit_ptr):
        """Device-buffer call (asynchronous): the three outputs are raw device pointers (int or
        None) on this context's GPU, e.g. torch tensors' data_ptr()."""
        seeing = np.ascontiguousarray(seeing, dtype=np.float64)
        gl = np.ascontiguousarray(gl, dtype=np.float64)
        l0 = np.ascontiguousarray(l0, dtype=np.float64)
        lbda = np.ascontiguousarray(lbda, dtype=np.float64)
        three = np.ascontiguousarray(three_lgs).astype(np.uint8)
        hh = np.ascontiguousarray(h, dtype=np.float64)
        mrec = mres = None
        if masks is not None:
            mrec = np.ascontiguousarray(masks[0]).astype(np.uint8).reshape(-1)
            mres = np.ascontiguousarray(masks[1]).astype(np.uint8).reshape(-1)
        _check(self.lib.mpsfr_reconstruct(
            self._h, seeing.size, _dptr(seeing), _dptr(gl), _dptr(l0), _u8ptr(three), _dptr(hh),
            float(wind_speed), int(npsflin), lbda.size, _dptr(lbda), _u8ptr(mrec), _u8ptr(mres),
            C.c_void_p(psf_ptr), C.c_void_p(sum_ptr), C.c_void_p(fit_ptr), 1))

    def simul_psd(self, seeing, gl, l0, three_lgs=False, h=(100, 10000), wind_speed=None, npsflin=1, masks=None):
        """simul_psd_wfm (psfrec.py:36-151): (npsflin^2, dim, dim) PSD, centred, reference units."""
        if wind_speed is None:
            wind_speed = float(np.full_like(np.array(h), 12.5)[0])      # psfrec.py:61
        hh = np.ascontiguousarray(h, dtype=np.float64)
        mrec = mres = None
        if masks is not None:
            mrec = np.ascontiguousarray(masks[0]).astype(np.uint8).reshape(-1)
            mres = np.ascontiguousarray(masks[1]).astype(np.uint8).reshape(-1)
        out = np.empty((int(npsflin) ** 2, self.dim, self.dim))
        _check(self.lib.mpsfr_simul_psd(self._h, float(seeing), float(gl), float(l0), int(bool(three_lgs)), _dptr(hh),
                                        float(wind_speed), int(npsflin), _u8ptr(mrec), _u8ptr(mres), _dptr(out)))
        return out

    def psf_from_psd(self, psd, lbda):
        """psf_muse (psfrec.py:644-686): PSD (dim, dim) or (ndir, dim, dim) -> (nl, 40, 40) stamps before the
        convolutions."""
        psd = np.ascontiguousarray(psd, dtype=np.float64)
        if psd.ndim == 2:
            psd = psd[None]
        if psd.shape[1:] != (self.dim, self.dim):
            raise ValueError('the PSD must be %d x %d for this context' % (self.dim, self.dim))
        lbda = np.ascontiguousarray(np.atleast_1d(lbda), dtype=np.float64)
        out = np.empty((lbda.size, self.dimpsf, self.dimpsf))
        _check(self.lib.mpsfr_psf_from_psd(self._h, psd.shape[0], _dptr(psd), lbda.size, _dptr(lbda), _dptr(out)))
        return out

    def convolve_stamps(self, lbda, seeing, gl, l0, psf):
        """convolve_final_psf (psfrec.py:874-930) on (ntask, nl, 40, 40) stamps (or (nl, 40, 40) for one task)."""
        lbda = np.ascontiguousarray(np.atleast_1d(lbda), dtype=np.float64)
        seeing = np.ascontiguousarray(np.atleast_1d(seeing), dtype=np.float64)
        gl = np.ascontiguousarray(np.atleast_1d(gl), dtype=np.float64)
        l0 = np.ascontiguousarray(np.atleast_1d(l0), dtype=np.float64)
        st = np.ascontiguousarray(psf, dtype=np.float64)
        single = st.ndim == 3
        st = st.reshape(seeing.size, lbda.size, self.dimpsf, self.dimpsf)
        out = np.empty_like(st)
        _check(self.lib.mpsfr_convolve_stamps(self._h, seeing.size, _dptr(seeing), _dptr(gl), _dptr(l0), lbda.size,
                                              _dptr(lbda), _dptr(st), _dptr(out)))
        return out[0] if single else out

    def fit_stamps(self, stamps):
        st = np.ascontiguousarray(stamps, dtype=np.float64).reshape(-1, self.dimpsf, self.dimpsf)
        out = np.empty((st.shape[0], NFIT))
        _check(self.lib.mpsfr_fit_stamps(self._h, st.shape[0], st.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p), 0))
        return out

    def debug_fetch(self, what, shape):
        out = np.empty(int(np.prod(shape)))
        n = _check(self.lib.mpsfr_debug_fetch(self._h, what.encode(), _dptr(out), out.size))
        if n != out.size:
            raise ValueError('%s: library returned %d values, expected %d' % (what, n, out.size))
        return out.reshape(shape)

    def profile(self):
        """{kernel name: (total_ms, launches)} accumulated since the last profile_reset()."""
        res = {}
        for i in range(self.lib.mpsfr_profile_count()):
            ms, n = C.c_double(), C.c_long()
            _check(self.lib.mpsfr_profile_get(self._h, i, C.byref(ms), C.byref(n)))
            res[self.lib.mpsfr_profile_name(i).decode()] = (ms.value, n.value)
        return res

    def profile_names(self):
        """Kernel names in id order (the ids `profile_only` takes)."""
        return [self.lib.mpsfr_profile_name(i).decode()
                for i in range(self.lib.mpsfr_profile_count())]

    def profile_reset(self):
        _check(self.lib.mpsfr_profile_reset(self._h))


class PendingResult:
    """An asynchronous host-output call (Context.reconstruct_async): the arrays of `wait()` are filled
    by the library when the ticket is waited for."""

    def __init__(self, ctx, ticket, arrays):
        self.ctx, self.ticket, self._arrays, self._done = ctx, ticket, arrays, False

    def wait(self):
        if not self._done:
            if self.ticket in self.ctx._pending:         # (not abandoned, not completed by a sync)
                _check(self.ctx.lib.mpsfr_wait(self.ctx._h, self.ticket))
                self.ctx._handed_over(self.ticket)
            elif self.ticket in self.ctx._abandoned:
                raise MpsfrError(-1, 'ticket %d was abandoned (Context.abandon): its results were dropped' % self.ticket)
            self._done = True
        return self._arrays


class PendingMulti:
    """An asynchronous multi-context call (Context.reconstruct_multi_async)."""

    def __init__(self, ctxs, handles, arrays):
        self.ctxs, self._handles, self._arrays, self._done = list(ctxs), handles, arrays, False

    def wait(self):
        if not self._done:
            try:
                _check(self.ctxs[0].lib.mpsfr_wait_multi(self._handles, len(self.ctxs)))
            finally:
                for c in self.ctxs:
                    for t in [t for t in c._pending if isinstance(t, tuple) and c._pending[t] is self._arrays]:
                        del c._pending[t]
            self._done = True
        return self._arrays


class ContextPool:
    """Independent batches pipelined through several contexts (each has its own HIP stream and
    workspaces): `next()` hands out the contexts in turn, so that consecutive `reconstruct_device`
    calls run concurrently on the GPU -- the transforms of one batch overlap the fit of the one
    before (two contexts: +24 % PSFs/s on the bench workload).  Each batch needs its own output
    buffers until its context has been synchronised."""

    def __init__(self, n=2, **kwargs):
        self.contexts = [Context(**kwargs) for _ in range(max(1, int(n)))]
        self._i = 0

    def next(self):
        c = self.contexts[self._i % len(self.contexts)]
        self._i += 1
        return c

    def set_option(self, key, value):
        for c in self.contexts:
            c.set_option(key, value)

    def sync(self):
        for c in self.contexts:
            c.sync()

    def close(self):
        for c in self.contexts:
            c.close()
        self.contexts = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

