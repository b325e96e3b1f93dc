"""ctypes binding of libzkr_hip.so (declarations follow include/zkr.h one to one)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKR_HIP_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "csrc", "libzkr_hip.so"))  # same override as index.js
PROOF_BYTES = 256
PARTIAL_BYTES = 640   # zkr.h ZKR_PARTIAL_BYTES
REPLICATE_MODES = {"auto": 0, "full": 1, "base": 2}   # zkr.h ZKR_REPLICATE_*
STAGES = ("ingest", "spmv", "ntt", "msm_sort", "msm_accum_g1", "msm_accum_g2", "msm_big", "msm_reduce", "total",
          "spmv_a", "ntt_pass", "combine_h")   # the last three: single streaming kernels (bench.py roofline.streaming)
_lib = None


class ZkrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("zkr error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load the HIP library; fail loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZkrError(-1, "HIP extension %s is missing: run __graft_entry__.build() "
                           "(make -C simple-zk-rollups_amd/csrc); there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, sz, u8p, i = c.c_void_p, c.c_size_t, c.c_char_p, c.c_int
    L.zkr_last_error.restype = c.c_char_p
    L.zkr_version.restype = c.c_char_p
    L.zkr_device_count.restype = i
    L.zkr_device_pci_bus_id.argtypes = [i, c.c_char_p, sz]
    L.zkr_key_load_websnark.argtypes = [u8p, sz, i, c.POINTER(vp)]
    L.zkr_key_free.argtypes = [vp]
    L.zkr_key_free.restype = None
    L.zkr_key_info.argtypes = [vp, c.POINTER(c.c_uint64)]
    L.zkr_key_save.argtypes = [vp, c.c_char_p]
    L.zkr_key_load_file.argtypes = [c.c_char_p, i, c.POINTER(vp)]
    L.zkr_key_slots.argtypes = [vp]
    L.zkr_key_fuse.argtypes = [vp]
    L.zkr_prove_batch_device.argtypes = [vp, c.POINTER(vp), sz, u8p, u8p, vp, u8p]
    L.zkr_key_windows.argtypes = [vp, c.POINTER(c.c_uint32), c.POINTER(c.c_uint32)]
    L.zkr_key_arena.argtypes = [vp, c.POINTER(vp), c.POINTER(sz)]
    L.zkr_key_adopt_arena.argtypes = [vp, sz, i, c.POINTER(vp)]
    L.zkr_key_base_arena.argtypes = [vp, c.POINTER(vp), c.POINTER(sz)]
    L.zkr_key_adopt_base_arena.argtypes = [vp, sz, i, c.POINTER(vp)]
    L.zkr_key_replicate.argtypes = [vp, i, i, c.POINTER(vp)]
    L.zkr_key_device.argtypes = [vp]
    L.zkr_key_replication.argtypes = [vp, c.POINTER(i), c.POINTER(i)]
    L.zkr_prove_sharded_last_form.argtypes = [c.POINTER(i), c.c_char_p, sz]
    L.zkr_prove_batch_multi.argtypes = [c.POINTER(vp), sz, c.POINTER(c.c_char_p), sz, sz, u8p, u8p, u8p]
    L.zkr_prove_batch_multi_device.argtypes = [c.POINTER(vp), sz, c.POINTER(vp), sz, u8p, u8p, u8p]
    L.zkr_key_shard.argtypes = [vp, c.c_uint, c.c_uint, i, c.POINTER(vp)]
    L.zkr_key_shard_info.argtypes = [vp, c.POINTER(c.c_uint32)]
    L.zkr_prove_partial.argtypes = [vp, u8p, sz, u8p]
    L.zkr_prove_partial_device.argtypes = [vp, vp, vp, u8p]
    L.zkr_prove_combine.argtypes = [vp, u8p, sz, u8p, u8p, u8p]
    L.zkr_prove_sharded.argtypes = [c.POINTER(vp), sz, u8p, sz, u8p, u8p, u8p]
    L.zkr_prove_sharded_device.argtypes = [c.POINTER(vp), sz, c.POINTER(vp), u8p, u8p, u8p]
    L.zkr_prove.argtypes = [vp, u8p, sz, u8p, u8p, u8p, vp]
    L.zkr_prove_device.argtypes = [vp, vp, u8p, u8p, u8p, vp]
    L.zkr_prove_submit.argtypes = [vp, vp, u8p, u8p, vp, c.POINTER(i)]
    L.zkr_prove_collect.argtypes = [vp, i, u8p]
    L.zkr_prove_batch.argtypes = [vp, c.POINTER(c.c_char_p), sz, sz, u8p, u8p, u8p]
    L.zkr_verify.argtypes = [u8p, sz, u8p, u8p, sz, c.POINTER(i)]
    L.zkr_verify_batch.argtypes = [u8p, sz, u8p, u8p, sz, sz, c.POINTER(i)]
    L.zkr_ntt.argtypes = [u8p, c.c_uint, i, i]
    L.zkr_msm_g1.argtypes = [u8p, u8p, sz, u8p, c.POINTER(i), i]
    L.zkr_msm_g2.argtypes = [u8p, u8p, sz, u8p, c.POINTER(i), i]
    L.zkr_calc_h.argtypes = [vp, u8p, sz, u8p]
    L.zkr_prof_enable.argtypes = [vp, i]
    L.zkr_prof_reset.argtypes = [vp]
    L.zkr_prof_get.argtypes = [vp, c.c_char_p, c.POINTER(c.c_double), c.POINTER(c.c_uint64)]
    L.zkr_synth_key.argtypes = [c.c_uint, c.c_uint, c.c_uint64, c.c_uint64, i, c.POINTER(vp), c.POINTER(vp), c.POINTER(sz),
                                c.POINTER(vp), c.POINTER(sz)]
    L.zkr_synth_websnark.argtypes = [c.c_uint, c.c_uint, c.c_uint64, c.c_uint64, i, c.POINTER(vp), c.POINTER(sz),
                                     c.POINTER(vp), c.POINTER(sz)]
    L.zkr_synth_witness.argtypes = [c.c_uint, c.c_uint, c.c_uint64, c.c_uint64, c.POINTER(vp), c.POINTER(sz)]
    L.zkr_setup_r1cs.argtypes = [u8p, sz, u8p, i, c.POINTER(vp), c.POINTER(vp), c.POINTER(sz)]
    L.zkr_setup_r1cs_websnark.argtypes = [u8p, sz, u8p, i, c.POINTER(vp), c.POINTER(sz), c.POINTER(vp), c.POINTER(sz)]
    L.zkr_synth_vk.argtypes = [vp, u8p, sz, c.POINTER(vp), c.POINTER(sz)]
    L.zkr_synth_set_shape.argtypes = [c.c_uint]
    L.zkr_free.argtypes = [vp]
    L.zkr_free.restype = None
    L.zkr_bench_fq_mul.argtypes = [i, c.POINTER(c.c_double)]
    L.zkr_bench_fq_mul_legacy.argtypes = [i, c.POINTER(c.c_double)]
    L.zkr_selftest_f29_forms.argtypes = [i, i, i, c.POINTER(c.c_uint32), sz, c.POINTER(c.c_uint32)]
    L.zkr_mimcsponge_multihash.argtypes = [u8p, sz, u8p]
    L.zkr_babyjub_pubkey.argtypes = [u8p, u8p]
    L.zkr_eddsa_sign.argtypes = [u8p, u8p, sz, u8p]
    L.zkr_eddsa_verify.argtypes = [u8p, sz, u8p, u8p, c.POINTER(i)]
    L.zkr_mimcsponge_multihash_batch.argtypes = [u8p, sz, c.c_uint, u8p, i]
    L.zkr_balance_tree_build.argtypes = [u8p, c.c_uint, u8p, i]
    L.zkr_babyjub_format_privkey.argtypes = [u8p, u8p]
    L.zkr_withdraw_r1cs.argtypes = [c.POINTER(vp), c.POINTER(sz)]
    L.zkr_withdraw_witness.argtypes = [u8p, u8p, c.POINTER(vp), c.POINTER(sz)]
    L.zkr_rollup_info.argtypes = [c.c_uint32, c.c_uint32, c.POINTER(c.c_uint32), c.POINTER(c.c_uint32), c.POINTER(c.c_uint32)]
    L.zkr_rollup_r1cs.argtypes = [c.c_uint32, c.c_uint32, c.POINTER(vp), c.POINTER(sz)]
    L.zkr_rollup_witness.argtypes = [c.c_uint32, c.c_uint32, u8p, sz, c.POINTER(vp), c.POINTER(sz)]
    L.zkr_rollup_witness_batch_device.argtypes = [c.c_uint32, c.c_uint32, u8p, sz, sz, vp, c.c_int]
    L.zkr_rollup_witness_program_host.argtypes = [c.c_uint32, c.c_uint32, u8p, sz, vp, c.POINTER(c.c_uint32), c.POINTER(c.c_uint32)]
    L.zkr_rollup_statement_text.argtypes = [c.c_uint32]
    L.zkr_rollup_statement_text.restype = c.c_char_p
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise ZkrError(rc, lib().zkr_last_error().decode())


def device_count():
    return lib().zkr_device_count()


def device_pci_bus_id(device=0) -> str:
    buf = ctypes.create_string_buffer(32)
    _check(lib().zkr_device_pci_bus_id(device, buf, 32))
    return buf.value.decode()


def version():
    return lib().zkr_version().decode()


def _take(ptr, n):
    data = ctypes.string_at(ptr, n)
    lib().zkr_free(ptr)
    return data


class ProvingKey:
    """Device-resident proving key (zkr_key*)."""

    def __init__(self, handle, device, keepalive=None):
        self._h = handle
        self.device = device
        self._keepalive = keepalive  # e.g. the torch tensor that owns an adopted arena

    @classmethod
    def load_websnark(cls, pk_bin: bytes, device=0):
        h = ctypes.c_void_p()
        _check(lib().zkr_key_load_websnark(bytes(pk_bin), len(pk_bin), device, ctypes.byref(h)))
        return cls(h, device)

    @classmethod
    def synth(cls, log_m, n_public=73, circuit_seed=0x5A4B0001, toxic_seed=0x5A4B00FF, device=0, want_aux=True):
        """-> (key, witness_bytes, aux_bytes_or_None); key points are computed on the GPU."""
        h, w, a = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        wl, al = ctypes.c_size_t(), ctypes.c_size_t()
        _check(lib().zkr_synth_key(log_m, n_public, circuit_seed, toxic_seed, device, ctypes.byref(h), ctypes.byref(w),
                                   ctypes.byref(wl), ctypes.byref(a) if want_aux else None,
                                   ctypes.byref(al) if want_aux else None))
        witness = _take(w, wl.value)
        aux = _take(a, al.value) if want_aux else None
        return cls(h, device), witness, aux

    @classmethod
    def adopt_arena(cls, dev_ptr, length, device, keepalive=None):
        h = ctypes.c_void_p()
        _check(lib().zkr_key_adopt_arena(ctypes.c_void_p(dev_ptr), length, device, ctypes.byref(h)))
        return cls(h, device, keepalive)

    @classmethod
    def adopt_base_arena(cls, dev_ptr, length, device):
        """Full key rebuilt on `device` from the compact arena bytes (zkr_key_adopt_base_arena); the buffer is not kept."""
        h = ctypes.c_void_p()
        _check(lib().zkr_key_adopt_base_arena(ctypes.c_void_p(dev_ptr), length, device, ctypes.byref(h)))
        return cls(h, device)

    @classmethod
    def setup_r1cs(cls, r1cs_bin: bytes, toxic=None, device=0):
        """Groth16 setup of an R1CS on the GPU (zkr_setup_r1cs; `snarkjs setup --protocol groth`).  toxic: None (OS
        CSPRNG) or five ints (t, alfa, beta, gamma, delta) for reproducible tests.  Returns (key, vk_bin)."""
        tb = None if toxic is None else b"".join(int(x).to_bytes(32, "little") for x in toxic)
        h, vk, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().zkr_setup_r1cs(bytes(r1cs_bin), len(r1cs_bin), tb, device, ctypes.byref(h), ctypes.byref(vk), ctypes.byref(n)))
        try:
            return cls(h, device), ctypes.string_at(vk, n.value)
        finally:
            lib().zkr_free(vk)

    @classmethod
    def load_file(cls, path, device=0):
        """Packed key written by save(): one read + one upload, no parse, no window-table rebuild."""
        h = ctypes.c_void_p()
        _check(lib().zkr_key_load_file(os.fsencode(path), device, ctypes.byref(h)))
        return cls(h, device)

    def save(self, path):
        _check(lib().zkr_key_save(self._h, os.fsencode(path)))

    def replicate(self, device, mode="auto"):
        """An independent replica of this key on `device` (zkr_key_replicate): device-to-device copy inside this process,
        no torch.distributed.  mode "full" (whole arena, adopted as is), "base" (compact arena + rebuild of the window
        levels) or "auto" (full when the devices address each other).  `device` may be the key's own."""
        h = ctypes.c_void_p()
        _check(lib().zkr_key_replicate(self._h, device, REPLICATE_MODES[mode], ctypes.byref(h)))
        return ProvingKey(h, device)

    def replication(self):
        """How this key came to its device (zkr_key_replication): {"mode": "none" | "full" | "base", "peer_direct": bool}."""
        mode, direct = ctypes.c_int(0), ctypes.c_int(0)
        _check(lib().zkr_key_replication(self._h, ctypes.byref(mode), ctypes.byref(direct)))
        return {"mode": {0: "none", 1: "full", 2: "base"}[mode.value], "peer_direct": bool(direct.value)}

    def shard(self, part, parts, device=None):
        """Shard `part` of `parts` of this (whole) key on `device` (default: the key's own): the points of one contiguous range
        of every MSM of a proof, all window levels, plus the whole QAP (zkr_key_shard)."""
        h = ctypes.c_void_p()
        dev = self.device if device is None else device
        _check(lib().zkr_key_shard(self._h, part, parts, dev, ctypes.byref(h)))
        return ProvingKey(h, dev)

    def bench_split_solo(self, d_witness_ptr) -> float:
        """Measurement only: this shard's share of a proof with a split calcH, alone, own buffers standing in for the other
        shards' (zkr_bench_shard_split_solo) -> milliseconds.  The result of the computation is meaningless and discarded."""
        ms = ctypes.c_double(0)
        _check(lib().zkr_bench_shard_split_solo(self._h, ctypes.c_void_p(d_witness_ptr), ctypes.byref(ms)))
        return ms.value

    def shard_info(self):
        out = (ctypes.c_uint32 * 6)()
        _check(lib().zkr_key_shard_info(self._h, out))
        return dict(zip(("part", "parts", "w_lo", "w_n", "h_lo", "h_n"), [int(x) for x in out]))

    def prove_partial(self, witness: bytes) -> bytes:
        """This shard's partial sums of A, B1, B2, C + H for one proof (zkr_prove_partial), PARTIAL_BYTES opaque bytes."""
        out = ctypes.create_string_buffer(PARTIAL_BYTES)
        _check(lib().zkr_prove_partial(self._h, bytes(witness), len(witness), out))
        return out.raw

    def prove_partial_device(self, d_witness_ptr, stream=None) -> bytes:
        out = ctypes.create_string_buffer(PARTIAL_BYTES)
        _check(lib().zkr_prove_partial_device(self._h, ctypes.c_void_p(d_witness_ptr), ctypes.c_void_p(stream or 0), out))
        return out.raw

    def prove_combine(self, partials, r=None, s=None) -> bytes:
        """The proof from the partial sums of all shards (zkr_prove_combine); self: any shard or the whole key."""
        out = ctypes.create_string_buffer(PROOF_BYTES)
        rb = None if r is None else int(r).to_bytes(32, "little")
        sb = None if s is None else int(s).to_bytes(32, "little")
        _check(lib().zkr_prove_combine(self._h, b"".join(partials), len(partials), rb, sb, out))
        return out.raw

    def close(self):
        if self._h:
            lib().zkr_key_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        out = (ctypes.c_uint64 * 10)()
        _check(lib().zkr_key_info(self._h, out))
        names = ("nVars", "nPublic", "domainSize", "nnzA", "nnzB", "ptsA", "ptsB1", "ptsB2", "ptsC", "ptsH")
        return dict(zip(names, [int(x) for x in out]))

    def arena(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().zkr_key_arena(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def base_arena(self):
        """(device pointer, length) of the compact arena: base points + QAP rows, no window levels (zkr_key_base_arena)."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().zkr_key_base_arena(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def prove(self, witness: bytes, r=None, s=None, stream=None) -> bytes:
        out = ctypes.create_string_buffer(PROOF_BYTES)
        rb = None if r is None else int(r).to_bytes(32, "little")
        sb = None if s is None else int(s).to_bytes(32, "little")
        _check(lib().zkr_prove(self._h, bytes(witness), len(witness), rb, sb, out, ctypes.c_void_p(stream or 0)))
        return out.raw

    def prove_device(self, d_witness_ptr, r=None, s=None, stream=None) -> bytes:
        out = ctypes.create_string_buffer(PROOF_BYTES)
        rb = None if r is None else int(r).to_bytes(32, "little")
        sb = None if s is None else int(s).to_bytes(32, "little")
        _check(lib().zkr_prove_device(self._h, ctypes.c_void_p(d_witness_ptr), rb, sb, out, ctypes.c_void_p(stream or 0)))
        return out.raw

    def synth_vk(self, aux: bytes) -> bytes:
        """vk_bin (zkr_verify layout) of a key made by ProvingKey.synth, from its checker blob."""
        out, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().zkr_synth_vk(self._h, bytes(aux), len(aux), ctypes.byref(out), ctypes.byref(n)))
        try:
            return ctypes.string_at(out, n.value)
        finally:
            lib().zkr_free(out)

    def windows(self):
        """{table: (window bits c, windows K)} for A, B1, B2, C, H."""
        cs, ks = (ctypes.c_uint32 * 5)(), (ctypes.c_uint32 * 5)()
        _check(lib().zkr_key_windows(self._h, cs, ks))
        return {t: (cs[i], ks[i]) for i, t in enumerate(("A", "B1", "B2", "C", "H"))}

    def prove_submit(self, d_witness_ptr, r=None, s=None, stream=None) -> int:
        """Enqueue the GPU side of one proof, return a ticket (zkr_prove_submit); at most two in flight."""
        rb = None if r is None else int(r).to_bytes(32, "little")
        sb = None if s is None else int(s).to_bytes(32, "little")
        ticket = ctypes.c_int(-1)
        _check(lib().zkr_prove_submit(self._h, ctypes.c_void_p(d_witness_ptr), rb, sb, ctypes.c_void_p(stream or 0), ctypes.byref(ticket)))
        return ticket.value

    def prove_collect(self, ticket) -> bytes:
        out = ctypes.create_string_buffer(PROOF_BYTES)
        _check(lib().zkr_prove_collect(self._h, int(ticket), out))
        return out.raw

    def prove_batch_device(self, d_witness_ptrs, rs=None, ss=None, stream=None, depth=None):
        """Independent proofs of one batch from witnesses resident in HBM (zkr_prove_batch_device): two submits in flight,
        proofs of small circuits fused into shared launches (fuse() per submit).  depth (or env ZKR_PIPELINE_DEPTH) selects
        the unfused submit / collect loop with that many single proofs in flight instead."""
        import collections
        import os
        depth = depth or int(os.environ.get("ZKR_PIPELINE_DEPTH", "0"))
        n = len(d_witness_ptrs)
        if not depth:
            if n == 0:
                return []
            arr = (ctypes.c_void_p * n)(*[ctypes.c_void_p(p) for p in d_witness_ptrs])
            rb, sb = _blinding_bytes(rs, ss, n)
            out = ctypes.create_string_buffer(256 * n)
            _check(lib().zkr_prove_batch_device(self._h, arr, n, rb, sb, ctypes.c_void_p(stream or 0), out))
            return [out.raw[256 * i:256 * i + 256] for i in range(n)]
        out, pending = [], collections.deque()
        for j, ptr in enumerate(d_witness_ptrs):
            if len(pending) == depth:
                out.append(self.prove_collect(pending.popleft()))
            pending.append(self.prove_submit(ptr, None if rs is None else rs[j], None if ss is None else ss[j], stream))
        while pending:
            out.append(self.prove_collect(pending.popleft()))
        return out

    def prove_batch(self, witnesses, rs=None, ss=None):
        """zkr_prove_batch: host witnesses (bytes, equal length) -> list of 256-byte proofs, uploads and proofs pipelined."""
        n = len(witnesses)
        if n == 0:
            return []
        wlen = _equal_length(witnesses)
        arr = (ctypes.c_char_p * n)(*[bytes(w) for w in witnesses])
        rb, sb = _blinding_bytes(rs, ss, n)
        out = ctypes.create_string_buffer(256 * n)
        _check(lib().zkr_prove_batch(self._h, arr, wlen, n, rb, sb, out))
        return [out.raw[256 * i:256 * i + 256] for i in range(n)]

    def slots(self):
        """Submits the key can hold in flight (zkr_prove_submit before zkr_prove_collect)."""
        return lib().zkr_key_slots(self._h)

    def fuse(self):
        """Proofs one batch submit runs in shared launches (zkr_key_fuse): 1 for circuits that fill the chip alone."""
        return lib().zkr_key_fuse(self._h)

    def calc_h(self, witness: bytes) -> bytes:
        m = self.info()["domainSize"]
        out = ctypes.create_string_buffer(32 * m)
        _check(lib().zkr_calc_h(self._h, bytes(witness), len(witness), out))
        return out.raw

    def prof_enable(self, on=True):
        _check(lib().zkr_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        _check(lib().zkr_prof_reset(self._h))

    def prof(self):
        """{stage: (ms_total, launches)} measured with hipEvents on the launch stream."""
        out = {}
        for st in STAGES:
            ms, n = ctypes.c_double(), ctypes.c_uint64()
            _check(lib().zkr_prof_get(self._h, st.encode(), ctypes.byref(ms), ctypes.byref(n)))
            out[st] = (ms.value, n.value)
        return out


def _blinding_bytes(rs, ss, n=None):
    """r / s of a batch as 32-byte LE records.  The C side reads 32 * count bytes of each: a short list must not reach it."""
    if (rs is None) != (ss is None):
        raise ValueError("pass both rs and ss or neither")
    if rs is not None and n is not None and (len(rs) != n or len(ss) != n):
        raise ValueError("a batch of %d proofs needs %d blinding scalars of each kind, got %d and %d" % (n, n, len(rs), len(ss)))
    rb = None if rs is None else b"".join(int(x).to_bytes(32, "little") for x in rs)
    sb = None if ss is None else b"".join(int(x).to_bytes(32, "little") for x in ss)
    return rb, sb


def _equal_length(witnesses):
    """The batch calls pass ONE witness_len for every pointer: a shorter buffer would be read past its end."""
    n0 = len(witnesses[0])
    for i, w in enumerate(witnesses):
        if len(w) != n0:
            raise ValueError("witness %d is %d bytes, witness 0 is %d: a batch takes witnesses of one circuit" % (i, len(w), n0))
    return n0


def prove_batch_multi(keys, witnesses, rs=None, ss=None):
    """zkr_prove_batch_multi: one batch of independent proofs over several replicas of a key (ProvingKey.replicate), proof i
    on keys[i mod len(keys)], one host thread per key inside the library.  witnesses: host bytes of equal length."""
    n = len(witnesses)
    if n == 0:
        return []
    ks = (ctypes.c_void_p * len(keys))(*[k._h for k in keys])
    wlen = _equal_length(witnesses)
    arr = (ctypes.c_char_p * n)(*[bytes(w) for w in witnesses])
    rb, sb = _blinding_bytes(rs, ss, n)
    out = ctypes.create_string_buffer(256 * n)
    _check(lib().zkr_prove_batch_multi(ks, len(keys), arr, wlen, n, rb, sb, out))
    return [out.raw[256 * i:256 * i + 256] for i in range(n)]


def prove_batch_multi_device(keys, d_witness_ptrs, rs=None, ss=None):
    """zkr_prove_batch_multi_device: the same for witnesses resident in HBM; d_witness_ptrs[i] lives on the device of
    keys[i mod len(keys)] and is complete (synchronise the producer first)."""
    n = len(d_witness_ptrs)
    if n == 0:
        return []
    ks = (ctypes.c_void_p * len(keys))(*[k._h for k in keys])
    arr = (ctypes.c_void_p * n)(*[ctypes.c_void_p(p) for p in d_witness_ptrs])
    rb, sb = _blinding_bytes(rs, ss, n)
    out = ctypes.create_string_buffer(256 * n)
    _check(lib().zkr_prove_batch_multi_device(ks, len(keys), arr, n, rb, sb, out))
    return [out.raw[256 * i:256 * i + 256] for i in range(n)]


def prove_sharded(shards, witness: bytes, r=None, s=None) -> bytes:
    """ONE proof over the shards of a key (ProvingKey.shard(i, parts) at position i), one host thread per shard inside the
    library, partial sums combined on the host (zkr_prove_sharded)."""
    ks = (ctypes.c_void_p * len(shards))(*[k._h for k in shards])
    out = ctypes.create_string_buffer(PROOF_BYTES)
    rb = None if r is None else int(r).to_bytes(32, "little")
    sb = None if s is None else int(s).to_bytes(32, "little")
    _check(lib().zkr_prove_sharded(ks, len(shards), bytes(witness), len(witness), rb, sb, out))
    return out.raw


def prove_sharded_device(shards, d_witness_ptrs, r=None, s=None) -> bytes:
    """The same with the full witness resident on every shard's device (d_witness_ptrs[i] on shards[i].device)."""
    ks = (ctypes.c_void_p * len(shards))(*[k._h for k in shards])
    arr = (ctypes.c_void_p * len(shards))(*[ctypes.c_void_p(p) for p in d_witness_ptrs])
    out = ctypes.create_string_buffer(PROOF_BYTES)
    rb = None if r is None else int(r).to_bytes(32, "little")
    sb = None if s is None else int(s).to_bytes(32, "little")
    _check(lib().zkr_prove_sharded_device(ks, len(shards), arr, rb, sb, out))
    return out.raw


def sharded_split_stats():
    """Of this thread's last sharded proof: None when every shard computed h for itself, else [part][phase] host milliseconds of
    the split calcH's phases (zkr_prove_sharded_split_stats)."""
    parts = ctypes.c_uint(0)
    ms = (ctypes.c_double * 64)()
    _check(lib().zkr_prove_sharded_split_stats(ctypes.byref(parts), ms))
    if parts.value == 0:
        return None
    return [[ms[8 * p + f] for f in range(5)] for p in range(parts.value)]


SHARDED_FORMS = {0: "none", 1: "split", 2: "replicated"}


def sharded_last_form():
    """Which form this thread's last sharded proof took and why (zkr_prove_sharded_last_form): {"form": "split" | "replicated" |
    "none", "reason": one line}.  A sharded proof that fell back to replicated calcH is otherwise only a slower number."""
    form = ctypes.c_int(0)
    buf = ctypes.create_string_buffer(256)
    _check(lib().zkr_prove_sharded_last_form(ctypes.byref(form), buf, 256))
    return {"form": SHARDED_FORMS.get(form.value, str(form.value)), "reason": buf.value.decode()}


def verify(vk_bin: bytes, proof: bytes, public_signals) -> bool:
    """zkr_verify: the Groth16 pairing check on the host (groth.isValid, common.ts:30-38).  public_signals: ints."""
    pub = b"".join(int(x).to_bytes(32, "little") for x in public_signals)
    ok = ctypes.c_int(0)
    _check(lib().zkr_verify(bytes(vk_bin), len(vk_bin), bytes(proof), pub, len(public_signals), ctypes.byref(ok)))
    return bool(ok.value)


def verify_batch(vk_bin: bytes, proofs, public_signals) -> bool:
    """zkr_verify_batch: True iff every proof verifies under the key (one merged pairing product, host only).
    proofs: list of 256-byte proofs; public_signals: one list of ints per proof."""
    n = len(proofs)
    if n == 0:
        return True
    n_pub = len(public_signals[0])
    assert len(public_signals) == n and all(len(p) == n_pub for p in public_signals)
    pb = b"".join(bytes(p) for p in proofs)
    sb = b"".join(int(v).to_bytes(32, "little") for p in public_signals for v in p)
    ok = ctypes.c_int()
    _check(lib().zkr_verify_batch(bytes(vk_bin), len(vk_bin), pb, sb, n, n_pub, ctypes.byref(ok)))
    return bool(ok.value)


def synth_set_shape(shape: int):
    """0 = rollup-shaped synthetic circuit (default), 1 = dense random (BASELINE configs[4])."""
    _check(lib().zkr_synth_set_shape(shape))


def ntt(data: bytes, inverse=False, device=0) -> bytes:
    n = len(data) // 32
    logn = n.bit_length() - 1
    if n < 2 or (1 << logn) != n:
        raise ValueError("length must be a power of two >= 2 elements")
    buf = ctypes.create_string_buffer(bytes(data), len(data))
    _check(lib().zkr_ntt(buf, logn, 1 if inverse else 0, device))
    return buf.raw


def _msm(fn, pb, points, scalars, device):
    n = len(scalars) // 32
    if len(points) != n * pb:
        raise ValueError("points/scalars length mismatch")
    out = ctypes.create_string_buffer(pb)
    inf = ctypes.c_int()
    _check(fn(bytes(points), bytes(scalars), n, out, ctypes.byref(inf), device))
    return None if inf.value else out.raw


def msm_g1(points: bytes, scalars: bytes, device=0):
    return _msm(lib().zkr_msm_g1, 64, points, scalars, device)


def msm_g2(points: bytes, scalars: bytes, device=0):
    return _msm(lib().zkr_msm_g2, 128, points, scalars, device)


def synth_websnark(log_m, n_public, circuit_seed, toxic_seed, device=0):
    """-> (provingKeyBin, witnessBin) in the binarify.ts layouts (host bytes; small sizes)."""
    p, w = ctypes.c_void_p(), ctypes.c_void_p()
    pl, wl = ctypes.c_size_t(), ctypes.c_size_t()
    _check(lib().zkr_synth_websnark(log_m, n_public, circuit_seed, toxic_seed, device, ctypes.byref(p), ctypes.byref(pl),
                                    ctypes.byref(w), ctypes.byref(wl)))
    return _take(p, pl.value), _take(w, wl.value)


def setup_r1cs_websnark(r1cs_bin: bytes, toxic=None, device=0):
    """zkr_setup_r1cs_websnark: Groth16 setup of an R1CS on the GPU, key returned as (provingKeyBin, vk_bin) -- the
    buffer `binarifyProvingKey` would write for the reference's unchanged caller."""
    tb = None if toxic is None else b"".join(int(x).to_bytes(32, "little") for x in toxic)
    pk, vk = ctypes.c_void_p(), ctypes.c_void_p()
    pl, vl = ctypes.c_size_t(), ctypes.c_size_t()
    _check(lib().zkr_setup_r1cs_websnark(bytes(r1cs_bin), len(r1cs_bin), tb, device, ctypes.byref(pk), ctypes.byref(pl), ctypes.byref(vk), ctypes.byref(vl)))
    return _take(pk, pl.value), _take(vk, vl.value)


def synth_witness(log_m, n_public, circuit_seed, witness_seed):
    """Another satisfying witness of the synthetic circuit (host only)."""
    w, wl = ctypes.c_void_p(), ctypes.c_size_t()
    _check(lib().zkr_synth_witness(log_m, n_public, circuit_seed, witness_seed, ctypes.byref(w), ctypes.byref(wl)))
    return _take(w, wl.value)


def selftest_f29_forms(field, form, records, device=0):
    """zkr_selftest_f29_forms: records = list of 8-tuples of 9-limb lists (raw 29-bit limbs); returns the 9 result limbs of
    each record as the device computes the product form (0: a b, 1: a^2, 2: a b + c d, 3: a b + c d + e f + g h)."""
    n = len(records)
    flat = (ctypes.c_uint32 * (72 * n))(*[l for rec in records for operand in rec for l in operand])
    out = (ctypes.c_uint32 * (9 * n))()
    _check(lib().zkr_selftest_f29_forms(device, field, form, flat, n, out))
    return [list(out[9 * k:9 * k + 9]) for k in range(n)]


def bench_fq_mul(device=0, legacy=False) -> float:
    """G Fq-mul/s of the hot path's multiplier (9 x 29-bit limbs); legacy=True: the 8 x 32-bit multiplier of field.hpp."""
    v = ctypes.c_double()
    _check((lib().zkr_bench_fq_mul_legacy if legacy else lib().zkr_bench_fq_mul)(device, ctypes.byref(v)))
    return v.value
