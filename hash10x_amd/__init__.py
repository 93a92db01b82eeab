"""hash10x_amd — MI355X (gfx950) implementation of hash10x's mosh-construction + clustering path.

This package is plumbing only: a ctypes binding of the C session layer (hash10x_amd/host/,
`libh10x_host.so`) which drives the HIP kernels through the C ABI of include/h10x.h
(`libh10x_hip.so`). There is no Python or CPU compute path: if the native libraries are missing or
no gfx950 device is present, construction fails loudly.

`Hash10x` mirrors the reference's command surface (hash10x.c:1200-1269):
    --readFQB -> read_fqb / read_fqb_file      --readHash  -> read_hash
    --writeHash -> write_hash                  --hashDepthRange -> depth_range
    --cluster -> cluster                       --clusterSplit -> cluster_split
with -k -w -r -B -N -c -ct as constructor / method arguments.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_HIP_SO = os.path.join(_HERE, "libh10x_hip.so")
_HOST_SO = os.path.join(_HERE, "libh10x_host.so")
ABI_VERSION = 3          # include/h10x.h H10X_ABI_VERSION this binding was written for (load_native checks the library's)


class Hash10xError(RuntimeError):
    """Raised with the reference's die() text where the reference would have died."""


class _Counters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in (
        "pairs", "kmers", "entries", "distinct", "clustered_codes", "sum_good", "sum_good_depth",
        "sum_hash_clustered", "fallback_blocks")] + [("cluster_class_counts", ctypes.c_uint64 * 4), ("cluster_first_mode", ctypes.c_uint64), ("cluster_overflow_blocks", ctypes.c_uint64), ("cluster_main", ctypes.c_uint64 * 4), ("cluster_phase_ticks", ctypes.c_uint64 * 8), ("list_words", ctypes.c_uint64 * 2), ("index_table_form", ctypes.c_uint64), ("shard_reply_path", ctypes.c_uint64)]


class _Sizes(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("hashNumber", ctypes.c_uint32), ("nBlocks", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("nClusHash", ctypes.c_uint64), ("nRecords", ctypes.c_uint64)]


class _ShardInfo(ctypes.Structure):
    _fields_ = [("rank", ctypes.c_int32), ("nranks", ctypes.c_int32), ("B", ctypes.c_int32), ("hashNumber", ctypes.c_uint32),
                ("nBlocksGlobal", ctypes.c_uint32), ("nSegs", ctypes.c_uint32), ("nEntriesGlobal", ctypes.c_uint64), ("nRecordsGlobal", ctypes.c_uint64)]


class _ShardSeg(ctypes.Structure):
    _fields_ = [("rank", ctypes.c_uint32), ("localStart", ctypes.c_uint32), ("count", ctypes.c_uint32), ("globalBase", ctypes.c_uint32),
                ("entries", ctypes.c_uint64), ("localEntryStart", ctypes.c_uint64), ("globalEntryStart", ctypes.c_uint64)]


_BLOCK_REP = np.dtype([("nGood", "<u4"), ("nClusHash", "<u4"), ("nClusRead", "<u4"), ("reserved", "<u4")])
_CLUSTER_REP = np.dtype([("n", "<u4"), ("nRead", "<u4"), ("nt", "<u4", (5,)), ("nBad", "<u4"), ("chr", "<i2"), ("pMin", "<u2"), ("pMax", "<u2"), ("nOtherListed", "<u2"), ("other", "<u4", (10,))])
assert _BLOCK_REP.itemsize == 16 and _CLUSTER_REP.itemsize == 80       # h10x_block_rep / h10x_cluster_rep (include/h10x.h)

_libs = None


def load_native():
    """Load libh10x_hip.so + libh10x_host.so (built in-tree by __graft_entry__.build()). No fallback."""
    global _libs
    if _libs is not None:
        return _libs
    for p in (_HIP_SO, _HOST_SO):
        if not os.path.exists(p):
            raise Hash10xError("native library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hash10x_amd has no CPU fallback)" % p)
    hip = ctypes.CDLL(_HIP_SO, mode=ctypes.RTLD_GLOBAL)
    host = ctypes.CDLL(_HOST_SO, mode=ctypes.RTLD_GLOBAL)
    vp, ci, cu64, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p
    host.h10x_session_new.restype = vp
    host.h10x_session_free.argtypes = [vp]
    host.h10x_session_set.argtypes = [vp, cs, ci]
    host.h10x_session_get.argtypes = [vp, cs]
    host.h10x_session_error.restype = cs
    host.h10x_session_error.argtypes = [vp]
    host.h10x_session_ctx.restype = vp
    host.h10x_session_ctx.argtypes = [vp]
    host.h10x_session_readFQB.argtypes = [vp, cs]
    host.h10x_session_begin.argtypes = [vp]
    host.h10x_session_after_read.argtypes = [vp]
    host.h10x_session_readFQB_mem.argtypes = [vp, vp, cu64]
    host.h10x_session_readFQB_dev.argtypes = [vp, vp, cu64]
    host.h10x_session_readHash.argtypes = [vp, cs]
    host.h10x_session_writeHash.argtypes = [vp, cs]
    host.h10x_session_hashDepthRange.argtypes = [vp, ci, ci]
    host.h10x_session_cluster.argtypes = [vp, ci, ci]
    host.h10x_session_clusterSplit.argtypes = [vp]
    host.h10x_host_array_dim.argtypes = [ci, ci, ctypes.c_int64]
    host.h10x_host_check_chunks.restype = ctypes.c_int64
    host.h10x_host_check_chunks.argtypes = [vp, cu64, ci, ci, cs, ci]
    host.h10x_host_partition.argtypes = [vp, cu64, ci, vp]
    host.h10x_session_shardReadFQB_mem.argtypes = [vp, vp, vp, cu64]
    host.h10x_session_shardReadFQB_dev.argtypes = [vp, vp, vp, cu64]
    host.h10x_session_shardGather.argtypes = [vp]
    host.h10x_session_shardReadHash.argtypes = [vp, vp, cs]
    host.h10x_session_shardReadFQB_file.argtypes = [vp, vp, cs, cu64, cu64]
    host.h10x_host_partition_file.argtypes = [cs, cu64, ci, vp, cs, ci]
    host.h10x_session_cribBuild.argtypes = [vp, cs, cs, vp, ci]
    host.h10x_session_clusterReport.argtypes = [vp, ci, ci, vp]
    host.h10x_session_cribSummary.argtypes = [vp, vp]
    host.h10x_session_hashStats.argtypes = [vp, vp]
    host.h10x_session_codeStats.argtypes = [vp, vp]
    hip.h10x_shard_info.argtypes = [vp, ctypes.POINTER(_ShardInfo)]
    hip.h10x_shard_segments.argtypes = [vp, vp, ctypes.c_uint32]
    hip.h10x_export_slice.argtypes = [vp, ci, cu64, cu64, vp]
    hip.h10x_shard_prepare_export.argtypes = [vp]
    hip.h10x_comm_unique_id.argtypes = [vp]
    hip.h10x_comm_create_rccl.argtypes = [ctypes.POINTER(vp), ci, ci, vp, ci, cs, ci]
    hip.h10x_comm_create_local.argtypes = [ctypes.POINTER(vp), ci]
    hip.h10x_comm_create_socket.argtypes = [ctypes.POINTER(vp), ci, ci, cs, ci, cs, ci]
    hip.h10x_comm_destroy.argtypes = [vp]
    hip.h10x_shard_barrier.argtypes = [vp]
    hip.h10x_shard_allreduce_max.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    hip.h10x_shard_allreduce_sum_u64.argtypes = [vp, vp, ctypes.c_uint32]
    hip.h10x_shard_allreduce_max_u64.argtypes = [vp, vp, ctypes.c_uint32]
    hip.h10x_ingest_fqb.argtypes = [vp, vp, cu64, ci]
    hip.h10x_ingest_reserve.argtypes = [vp, cu64]
    hip.h10x_build_id.restype = cs
    hip.h10x_device_malloc.restype = vp
    hip.h10x_device_malloc.argtypes = [ci, cu64]
    hip.h10x_device_free.argtypes = [ci, vp]
    hip.h10x_device_upload.argtypes = [ci, vp, vp, cu64]
    hip.h10x_device_synchronize.argtypes = [ci]
    hip.h10x_device_mem_info.argtypes = [ci, ctypes.POINTER(cu64), ctypes.POINTER(cu64)]
    hip.h10x_crib_genome.argtypes = [vp, vp, vp, ctypes.c_uint32, ci, ctypes.POINTER(cu64), ctypes.POINTER(cu64)]
    hip.h10x_crib_finish.argtypes = [vp]
    hip.h10x_cluster_report.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp, cu64, ctypes.POINTER(cu64)]
    hip.h10x_device_count.restype = ci
    hip.h10x_warm.restype = ci; hip.h10x_warm.argtypes = [ci]
    hip.h10x_alloc_stats.restype = None; hip.h10x_alloc_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    hip.h10x_abi_version.restype = ci
    hip.h10x_factor1_from_seed.restype = cu64
    hip.h10x_factor1_from_seed.argtypes = [ctypes.c_int32]
    hip.h10x_timing_enable.argtypes = [vp, ci]
    hip.h10x_timing_count.argtypes = [vp]
    hip.h10x_timing_name.restype = cs
    hip.h10x_timing_name.argtypes = [vp, ci]
    hip.h10x_timing_get.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(cu64)]
    hip.h10x_timing_reset.argtypes = [vp]
    hip.h10x_exchange_name.restype = cs; hip.h10x_exchange_name.argtypes = [ci]
    hip.h10x_exchange_beside.restype = ci; hip.h10x_exchange_beside.argtypes = [vp, ci]
    hip.h10x_timing_wait_get.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double)]
    hip.h10x_exchange_get.argtypes = [vp, ci, ctypes.POINTER(cu64), ctypes.POINTER(cu64), ctypes.POINTER(cu64), ctypes.POINTER(cu64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    hip.h10x_comm_local_serialize.argtypes = [vp, ci]
    hip.h10x_comm_turn_begin.argtypes = [vp]
    hip.h10x_comm_turn_end.argtypes = [vp, ci]
    hip.h10x_get_counters.argtypes = [vp, ctypes.POINTER(_Counters)]
    hip.h10x_get_sizes.argtypes = [vp, ctypes.POINTER(_Sizes)]
    hip.h10x_set_option.argtypes = [vp, cs, ctypes.c_int64]
    hip.h10x_last_error.restype = cs
    hip.h10x_last_error.argtypes = [vp]
    hip.h10x_export.argtypes = [vp, vp, vp, vp, vp, vp]
    if hip.h10x_abi_version() != ABI_VERSION:
        raise RuntimeError("libh10x_hip.so speaks ABI %d, hash10x_amd/__init__.py was written for %d (include/h10x.h H10X_ABI_VERSION): rebuild with "
                           "`python -c 'import __graft_entry__ as g; g.build()'`" % (hip.h10x_abi_version(), ABI_VERSION))
    _libs = (hip, host)
    return _libs


def device_count():
    return load_native()[0].h10x_device_count()


def device_mem_info(device=0):
    """(free, total) bytes of the device's memory as the driver reports them"""
    hip, _ = load_native()
    f, t = ctypes.c_uint64(0), ctypes.c_uint64(0)
    if hip.h10x_device_mem_info(int(device), ctypes.byref(f), ctypes.byref(t)):
        raise Hash10xError("h10x_device_mem_info failed")
    return f.value, t.value


def warm(device=0):
    """load the library's device code on `device` ahead of its first use (h10x_warm): 0 on success"""
    return load_native()[0].h10x_warm(int(device))


def alloc_stats():
    """(blocks, bytes) this process has obtained from hipMalloc so far (h10x_alloc_stats)"""
    a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
    load_native()[0].h10x_alloc_stats(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def build_id():
    """what libh10x_hip.so was built from: "src:<sha256 over its sources and compile flags, 16 hex>" (embedded at build time by csrc/Makefile)"""
    return load_native()[0].h10x_build_id().decode()


class DeviceRecords:
    """A sorted .fqb image resident in HBM (hipMalloc through the library, no torch involved)."""

    def __init__(self, records, device=0):
        hip = load_native()[0]
        r = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1)
        self.n_records, self.device, self._hip = r.size // 30, device, hip
        self.ptr = hip.h10x_device_malloc(device, r.nbytes)
        if not self.ptr:
            raise Hash10xError("hipMalloc of %d bytes failed on device %d" % (r.nbytes, device))
        if hip.h10x_device_upload(device, self.ptr, r.ctypes.data, r.nbytes):
            raise Hash10xError("upload to device %d failed" % device)

    def free(self):
        if getattr(self, "ptr", None):
            self._hip.h10x_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synchronize(device=0):
    load_native()[0].h10x_device_synchronize(device)


class Comm:
    """Communicator of the sharded path: RCCL (one process per GPU) or in-process (N ranks = N threads)."""

    def __init__(self, handle, rank, size):
        self.handle, self.rank, self.size = handle, rank, size

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        if load_native()[0].h10x_comm_unique_id(buf):
            raise Hash10xError("ncclGetUniqueId failed")
        return buf.raw

    @staticmethod
    def rccl(rank, size, unique_id, device):
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        if load_native()[0].h10x_comm_create_rccl(ctypes.byref(h), rank, size, unique_id, device, err, 512):
            raise Hash10xError(err.value.decode())
        return Comm(h, rank, size)

    @staticmethod
    def socket(rank, size, addr="127.0.0.1", base_port=29700):
        """One process per rank, host-staged over TCP: the multi-process path where RCCL cannot run (ranks sharing a GPU)."""
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        if load_native()[0].h10x_comm_create_socket(ctypes.byref(h), rank, size, addr.encode(), int(base_port), err, 512):
            raise Hash10xError(err.value.decode())
        return Comm(h, rank, size)

    @staticmethod
    def local(size):
        arr = (ctypes.c_void_p * size)()
        load_native()[0].h10x_comm_create_local(arr, size)
        return [Comm(ctypes.c_void_p(arr[i]), i, size) for i in range(size)]

    def serialize(self, on=True):
        """in-process communicators: the ranks take turns on the device they share (h10x_comm_local_serialize); bracket every command with turn_begin / turn_end"""
        if load_native()[0].h10x_comm_local_serialize(self.handle, 1 if on else 0):
            raise Hash10xError("not an in-process communicator")

    def turn_begin(self):
        load_native()[0].h10x_comm_turn_begin(self.handle)

    def turn_end(self, device=0):
        load_native()[0].h10x_comm_turn_end(self.handle, int(device))

    def destroy(self):
        if self.handle:
            load_native()[0].h10x_comm_destroy(self.handle)
            self.handle = None


def partition(records, parts):
    """Record index cuts of contiguous barcode-range shards (h10x_host_partition)."""
    r = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1)
    cut = (ctypes.c_uint64 * (parts + 1))()
    if load_native()[1].h10x_host_partition(r.ctypes.data, r.size // 30, parts, cut):
        raise Hash10xError("partition failed")
    return [int(x) for x in cut]


class Hash10x:
    """One hash10x session on one GPU (the reference's process-global state, hash10x.c:85-104)."""

    def __init__(self, k=21, w=31, r=17, B=28, device=0):
        self._hip, self._host = load_native()
        self._s = self._host.h10x_session_new()
        if not self._s:
            raise Hash10xError("out of memory")
        for n, v in (("k", k), ("w", w), ("r", r), ("B", B), ("device", device)):
            self._host.h10x_session_set(self._s, n.encode(), int(v))
        self._timing = False

    def close(self):
        if getattr(self, "_s", None):
            self._host.h10x_session_free(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise Hash10xError(self._host.h10x_session_error(self._s).decode())

    def _ctx(self):
        return self._host.h10x_session_ctx(self._s)

    def _after_init(self):
        ctx = self._ctx()
        if ctx and self._timing:
            self._hip.h10x_timing_enable(ctx, 1)

    # ---- commands ------------------------------------------------------------------------------
    def _pre(self, N, chunk):
        self._host.h10x_session_set(self._s, b"N", int(N))
        self._host.h10x_session_set(self._s, b"c", int(chunk))

    def read_fqb(self, records, N=0, chunk=100000):
        """--readFQB on an in-memory image of the sorted .fqb file (30 uint32 per read pair)."""
        r = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1)
        if r.size % 30:
            r = r[: r.size - r.size % 30]
        self._pre(N, chunk)
        self._chk(self._host.h10x_session_readFQB_mem(self._s, r.ctypes.data, r.size // 30))
        self._after_init()

    def ingest_fqb(self, chunks, N=0, chunk=100000, reserve=0):
        """--readFQB through the streaming C ABI (h10x_ingest_fqb): `chunks` = an iterable of record arrays (any sizes), appended on the
        device one by one; the last call closes the ingest. The session's parameters are latched as for read_fqb."""
        self._pre(N, chunk)
        self._chk(self._host.h10x_session_begin(self._s))
        if self._hip.h10x_set_option(self._ctx(), b"chunk_size", int(chunk)):
            raise Hash10xError(self._hip.h10x_last_error(self._ctx()).decode())
        if reserve:
            self._chk_ctx(self._hip.h10x_ingest_reserve(self._ctx(), int(reserve)))
        fed, cut = 0, False                                  # -N n: the first n records, and then no end-of-file pass (hash10x.c:202-208), as read_fqb does it
        for part in chunks:
            r = np.ascontiguousarray(part, dtype=np.uint32).reshape(-1)
            take = r.size // 30
            if N > 0 and fed + take >= N:
                take, cut = N - fed, True
            if take:
                self._chk_ctx(self._hip.h10x_ingest_fqb(self._ctx(), r.ctypes.data, take, 0))
            fed += take
            if cut:
                break
        if self._hip.h10x_set_option(self._ctx(), b"chunk_eof_pass", 0 if cut else 1):
            raise Hash10xError(self._hip.h10x_last_error(self._ctx()).decode())
        self._chk_ctx(self._hip.h10x_ingest_fqb(self._ctx(), None, 0, 1))
        self._chk(self._host.h10x_session_after_read(self._s))
        self._after_init()

    def read_fqb_file(self, path, N=0, chunk=100000):
        self._pre(N, chunk)
        self._chk(self._host.h10x_session_readFQB(self._s, os.fsencode(path)))
        self._after_init()

    def read_fqb_device(self, dev_ptr, n_records, N=0):
        """--readFQB with the records already resident in HBM (dev_ptr = device address)."""
        self._pre(N, 100000)
        self._chk(self._host.h10x_session_readFQB_dev(self._s, ctypes.c_void_p(dev_ptr), int(n_records)))
        self._after_init()

    def shard_read_fqb(self, comm, shard_records):
        """Sharded --readFQB: this rank's barcode range (host image); collective over comm."""
        r = np.ascontiguousarray(shard_records, dtype=np.uint32).reshape(-1)
        self._chk(self._host.h10x_session_shardReadFQB_mem(self._s, comm.handle, r.ctypes.data, r.size // 30))
        self._after_init()

    def shard_read_fqb_device(self, comm, dev_ptr, n_records):
        self._chk(self._host.h10x_session_shardReadFQB_dev(self._s, comm.handle, ctypes.c_void_p(dev_ptr), int(n_records)))
        self._after_init()

    def shard_read_hash(self, comm, path):
        """Sharded --readHash (collective): every rank loads the replicated tables and its own cut of the file's blocks."""
        self._chk(self._host.h10x_session_shardReadHash(self._s, comm.handle, os.fsencode(path)))
        self._after_init()

    def shard_gather(self):
        """Collective: rank 0 ends up with the whole state (then write_hash there)."""
        self._chk(self._host.h10x_session_shardGather(self._s))

    def shard_read_fqb_file(self, comm, path, first, n, chunk=100000):
        """Sharded --readFQB from a file: this rank's records [first, first + n), streamed into HBM; the reference's chunk
        semantics (-c) are applied over the whole file."""
        self._host.h10x_session_set(self._s, b"c", int(chunk))
        self._chk(self._host.h10x_session_shardReadFQB_file(self._s, comm.handle, os.fsencode(path), int(first), int(n)))
        self._after_init()

    def _chk_ctx(self, rc):
        if rc != 0:
            raise Hash10xError(self._hip.h10x_last_error(self._ctx()).decode())

    def shard_barrier(self):
        self._chk_ctx(self._hip.h10x_shard_barrier(self._ctx()))

    def shard_allreduce_max(self, value):
        v = ctypes.c_double(value)
        self._chk_ctx(self._hip.h10x_shard_allreduce_max(self._ctx(), ctypes.byref(v)))
        return v.value

    def shard_allreduce_sum_u64(self, values):
        """sums (mod 2^64) over the ranks of a small list of integers; the identity on an unsharded context"""
        v = np.array([int(x) & 0xFFFFFFFFFFFFFFFF for x in values], dtype=np.uint64)
        self._chk_ctx(self._hip.h10x_shard_allreduce_sum_u64(self._ctx(), v.ctypes.data, len(v)))
        return [int(x) for x in v]

    def shard_allreduce_max_u64(self, values):
        """maxima over the ranks of a small list of integers; the identity on an unsharded context"""
        v = np.array([int(x) & 0xFFFFFFFFFFFFFFFF for x in values], dtype=np.uint64)
        self._chk_ctx(self._hip.h10x_shard_allreduce_max_u64(self._ctx(), v.ctypes.data, len(v)))
        return [int(x) for x in v]

    def export_slice(self, table, first, count):
        """elements [first, first + count) of one table of THIS rank (h10x_export_slice; works on shards): 3 = blocks (32 B), 4 = ClusterHash (8 B)"""
        width = {0: 4, 1: 8, 2: 4, 3: 32, 4: 8, 5: 4}[table]
        out = np.zeros(max(int(count), 1) * width, dtype=np.uint8)
        self._chk_ctx(self._hip.h10x_export_slice(self._ctx(), table, int(first), int(count), out.ctypes.data))
        return out[: int(count) * width]

    def shard_info(self):
        z = _ShardInfo()
        self._chk_ctx(self._hip.h10x_shard_info(self._ctx(), ctypes.byref(z)))
        return {n: int(getattr(z, n)) for n, _ in _ShardInfo._fields_}

    def shard_segments(self):
        n = self.shard_info()["nSegs"]
        arr = (_ShardSeg * (n + 1))()
        self._chk_ctx(self._hip.h10x_shard_segments(self._ctx(), arr, n + 1))
        return [{f: int(getattr(arr[i], f)) for f, _ in _ShardSeg._fields_} for i in range(n)]

    # ---- text commands: `out` = a path (this rank prints there) or None (take part only) ----------------
    def _with_file(self, out, fn):
        libc = ctypes.CDLL(None)
        libc.fopen.restype = ctypes.c_void_p
        libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        libc.fclose.argtypes = [ctypes.c_void_p]
        f = libc.fopen(os.fsencode(out), b"a") if out else None
        try:
            self._chk(fn(ctypes.c_void_p(f) if f else None))
        finally:
            if f:
                libc.fclose(f)

    def crib_build(self, fa1, fa2, out=None, tables=False):
        self._with_file(out, lambda f: self._host.h10x_session_cribBuild(self._s, os.fsencode(fa1), os.fsencode(fa2), f, 1 if tables else 0))

    def crib_genomes(self, codes_a, codes_b, piece=60000000):
        """--cribBuild through the C ABI without FASTA files (h10x_crib_genome x 2 + h10x_crib_finish, hash10x.c:426-494): each haplotype as base codes 0..3, cut into
        sequences of `piece` bases as the truth FASTAs of gen_fqb are. Returns [(known, unknown) moshes of genome 1, of genome 2]."""
        out = []
        for which, codes in enumerate((codes_a, codes_b)):
            codes = np.ascontiguousarray(codes, dtype=np.uint8)
            starts = np.arange(0, codes.size + piece, piece, dtype=np.uint64)
            starts[-1] = codes.size
            if starts.size >= 2 and starts[-2] >= codes.size:
                starts = starts[:-1]; starts[-1] = codes.size
            known, unknown = ctypes.c_uint64(0), ctypes.c_uint64(0)
            self._chk_ctx(self._hip.h10x_crib_genome(self._ctx(), codes.ctypes.data, starts.ctypes.data, int(starts.size - 1), which, ctypes.byref(known), ctypes.byref(unknown)))
            out.append((known.value, unknown.value))
        self._chk_ctx(self._hip.h10x_crib_finish(self._ctx()))
        return out

    def cluster_report_figures(self, first_block=1, n_blocks=None, run=1 << 16):
        """codeClusterReport's figures (hash10x.c:870-952) reduced over blocks [first, first + n) of THIS rank without forming the text: the sums tests/orc.report_digest reads
        off the reference's CODE_CLUSTER lines — clusters printed (n > 0), those without an OTHER list, those with a location, the sums of their spans, reads and hashes —
        plus the sum of nGoodHash and of nClusHash over the blocks."""
        z = self.sizes()
        if n_blocks is None:
            n_blocks = z["nBlocks"] - first_block
        tot = dict(clusters=0, clusters_without_OTHER=0, clusters_located=0, sum_span=0, sum_reads=0, sum_hashes=0, sum_nGood=0, sum_nClusHash=0)
        brep = np.zeros(run, dtype=_BLOCK_REP)
        crep = np.zeros(run * 16, dtype=_CLUSTER_REP)
        at = first_block
        while at < first_block + n_blocks:
            nb = min(run, first_block + n_blocks - at)
            ncl = ctypes.c_uint64(0)
            rc = self._hip.h10x_cluster_report(self._ctx(), int(at), int(nb), brep.ctypes.data, crep.ctypes.data, ctypes.c_uint64(crep.size), ctypes.byref(ncl))
            if rc and ncl.value > crep.size:                 # more clusters than the buffer holds: the call says how many
                crep = np.zeros(int(ncl.value) + 1024, dtype=_CLUSTER_REP)
                rc = self._hip.h10x_cluster_report(self._ctx(), int(at), int(nb), brep.ctypes.data, crep.ctypes.data, ctypes.c_uint64(crep.size), ctypes.byref(ncl))
            self._chk_ctx(rc)
            c = crep[: ncl.value]
            c = c[c["n"] > 0]
            loc = c["chr"] != 0
            tot["clusters"] += int(c.size); tot["clusters_without_OTHER"] += int((c["nBad"] == 0).sum()); tot["clusters_located"] += int(loc.sum())
            tot["sum_span"] += int((c["pMax"][loc].astype(np.int64) - c["pMin"][loc].astype(np.int64) + 1).sum())
            tot["sum_reads"] += int(c["nRead"].sum(dtype=np.int64)); tot["sum_hashes"] += int(c["n"].sum(dtype=np.int64))
            tot["sum_nGood"] += int(brep["nGood"][:nb].sum(dtype=np.int64)); tot["sum_nClusHash"] += int(brep["nClusHash"][:nb].sum(dtype=np.int64))
            at += nb
        return tot

    def cluster_report(self, code_min, code_max, out=None):
        self._with_file(out, lambda f: self._host.h10x_session_clusterReport(self._s, int(code_min), int(code_max), f))

    def crib_summary(self, out=None):
        self._with_file(out, lambda f: self._host.h10x_session_cribSummary(self._s, f))

    def hash_stats(self, out=None):
        self._with_file(out, lambda f: self._host.h10x_session_hashStats(self._s, f))

    def code_stats(self, out=None):
        self._with_file(out, lambda f: self._host.h10x_session_codeStats(self._s, f))

    def read_hash(self, path):
        self._chk(self._host.h10x_session_readHash(self._s, os.fsencode(path)))
        self._after_init()

    def write_hash(self, path):
        self._chk(self._host.h10x_session_writeHash(self._s, os.fsencode(path)))

    def depth_range(self, lo, hi):
        self._chk(self._host.h10x_session_hashDepthRange(self._s, int(lo), int(hi)))

    def cluster(self, code_min=1, code_max=0, threshold=5):
        self._host.h10x_session_set(self._s, b"ct", int(threshold))
        self._chk(self._host.h10x_session_cluster(self._s, int(code_min), int(code_max)))

    def cluster_split(self):
        self._chk(self._host.h10x_session_clusterSplit(self._s))

    # ---- measurement / test hooks ---------------------------------------------------------------
    def enable_timing(self, on=True):
        self._timing = bool(on)
        self._host.h10x_session_set(self._s, b"timing", 1 if on else 0)
        if self._ctx():
            self._hip.h10x_timing_enable(self._ctx(), 1 if on else 0)

    def set_option(self, name, value):
        """Test knobs (e.g. stage_a_max_slots) forwarded to the context of the next read_fqb/read_hash."""
        if self._host.h10x_session_set(self._s, name.encode(), int(value)):
            raise Hash10xError(self._host.h10x_session_error(self._s).decode())

    def timings(self):
        ctx = self._ctx()
        out = {}
        if not ctx:
            return out
        for i in range(self._hip.h10x_timing_count(ctx)):
            ms, n = ctypes.c_double(), ctypes.c_uint64()
            self._hip.h10x_timing_get(ctx, i, ctypes.byref(ms), ctypes.byref(n))
            out[self._hip.h10x_timing_name(ctx, i).decode()] = (ms.value, n.value)
        return out

    def stage_waits(self):
        """per stage timer: the ms of it spent inside exchanges (sharded contexts)"""
        out = {}
        for i in range(self._hip.h10x_timing_count(self._ctx())):
            ms = ctypes.c_double(0)
            self._hip.h10x_timing_wait_get(self._ctx(), i, ctypes.byref(ms))
            if ms.value:
                out[self._hip.h10x_timing_name(self._ctx(), i).decode()] = ms.value
        return out

    def exchanges(self):
        """per kind of collective of the sharded path: calls, bytes to / from other ranks, bytes the busiest peer got, ms (waits included) and the part of it inside stage timers"""
        out = {}
        for i in range(self._hip.h10x_exchange_count()):
            v = [ctypes.c_uint64(0) for _ in range(4)]; ms, ms_in = ctypes.c_double(0), ctypes.c_double(0)
            self._hip.h10x_exchange_get(self._ctx(), i, ctypes.byref(v[0]), ctypes.byref(v[1]), ctypes.byref(v[2]), ctypes.byref(v[3]), ctypes.byref(ms), ctypes.byref(ms_in))
            if v[0].value:
                b = self._hip.h10x_exchange_beside(self._ctx(), i)
                out[self._hip.h10x_exchange_name(i).decode()] = {"calls": v[0].value, "bytes_out": v[1].value, "bytes_in": v[2].value, "max_peer_out": v[3].value, "ms": ms.value, "ms_in_stages": ms_in.value,
                                                                 "beside": self._hip.h10x_timing_name(self._ctx(), b).decode() if b >= 0 else None}
        return out

    def reset_timings(self):
        if self._ctx():
            self._hip.h10x_timing_reset(self._ctx())

    def counters(self):
        c = _Counters()
        if self._ctx():
            self._hip.h10x_get_counters(self._ctx(), ctypes.byref(c))
        out = {n: int(getattr(c, n)) for n, _ in _Counters._fields_ if n not in ("cluster_phase_ticks", "cluster_class_counts", "cluster_main", "list_words")}
        out["list_words"] = [int(x) for x in c.list_words]
        out["cluster_main"] = [int(x) for x in c.cluster_main]
        out["cluster_class_counts"] = [int(x) for x in c.cluster_class_counts]
        out["cluster_phase_ticks"] = [int(x) for x in c.cluster_phase_ticks]
        return out

    def sizes(self):
        z = _Sizes()
        if not self._ctx() or self._hip.h10x_get_sizes(self._ctx(), ctypes.byref(z)):
            raise Hash10xError("no hash state loaded")
        return {n: int(getattr(z, n)) for n, _ in _Sizes._fields_ if n != "reserved"}

    def export_blocks(self):
        """nBlocks ClusterBlock records as a structured array (for tests/bench sanity checks)."""
        z = self.sizes()
        dt = np.dtype([("nRead", "<u4"), ("nHash", "<u4"), ("nSubCluster", "<u4"), ("clusterParent", "<u4"),
                       ("ptr", "<u8"), ("pointToMin", "<f8")])
        b = np.zeros(z["nBlocks"], dtype=dt)
        if self._hip.h10x_export(self._ctx(), None, None, None, b.ctypes.data, None):
            raise Hash10xError(self._hip.h10x_last_error(self._ctx()).decode())
        return b

    def export_clushash(self):
        """All ClusterHash records (blocks 1.. concatenated) as a structured array."""
        z = self.sizes()
        dt = np.dtype([("hash", "<u4"), ("read", "<u2"), ("subCluster", "u1"), ("flags", "u1")])
        ch = np.zeros(max(z["nClusHash"], 1), dtype=dt)
        if self._hip.h10x_export(self._ctx(), None, None, None, None, ch.ctypes.data):
            raise Hash10xError(self._hip.h10x_last_error(self._ctx()).decode())
        return ch[: z["nClusHash"]]

    def export_depth(self):
        """hashDepth[0 .. hashNumber) as uint32."""
        z = self.sizes()
        d = np.zeros(z["hashNumber"], dtype=np.uint32)
        if self._hip.h10x_export(self._ctx(), None, None, d.ctypes.data, None, None):
            raise Hash10xError(self._hip.h10x_last_error(self._ctx()).decode())
        return d
