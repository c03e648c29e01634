"""Owner-partitioned spectrum build, one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Replaces the exchange of DistributedKmerSpectrum::_buildKmerSpectrumMPI
(src/DistributedFunctions.h:340-458): every rank extracts the good k-mers of its own
reads, bins them by owner = ((lookup3 >> 24) & 0x7ffff) % world (src/Kmer.h:2284-2295),
and one all-to-all per chunk moves each record to its owner (MPI_Alltoallv of
src/MPIBuffer.h:588-600; the 'int dataSize' prefix becomes the counts all-to-all).
Termination is implicit: every rank runs the same number of chunks.
"""
import torch
import torch.distributed as dist


def exchange_records(records, seg_counts, seg_capacity, rec_bytes, group=None):
    """records: uint8 tensor [world * seg_capacity * rec_bytes], segment s holds
    seg_counts[s] records for rank s.  Returns (recv uint8 tensor, n_records).

    The payload travels as int32 rows of one record each, so the split sizes handed to the
    collective count records (a byte count overflows 32 bits at ~1.8e8 twelve-byte records)."""
    world = dist.get_world_size(group)
    send_counts = seg_counts.to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc = [int(x) for x in send_counts.cpu().tolist()]
    rc = [int(x) for x in recv_counts.cpu().tolist()]
    if max(sc) > seg_capacity:
        raise RuntimeError("owner segment overflow: %d > %d" % (max(sc), seg_capacity))
    assert rec_bytes % 4 == 0
    words = rec_bytes // 4
    rows = records.view(torch.int32).view(world, seg_capacity, words)
    recv = _all_to_all_rows(rows, sc, rc, group)
    return recv.view(torch.uint8).view(-1), sum(rc)


def _all_to_all_rows(rows, sc, rc, group=None):
    """rows: int32 [world, seg_cap, words]; rank s gets rows[s, :sc[s]].  Returns the received rows [sum(rc), words].
    RCCL takes the owner segments as they lie (a list of views: grouped send/recv, no staging copy of the payload);
    gloo only has the single-tensor form, so the CPU tests go through one concatenation."""
    world = rows.shape[0]
    recv = torch.empty((sum(rc), rows.shape[2]), dtype=rows.dtype, device=rows.device)
    if dist.get_backend(group) == "nccl":
        dist.all_to_all(list(recv.split(rc)), [rows[s, :sc[s]] for s in range(world)], group=group)
    else:
        send = torch.cat([rows[s, :sc[s]] for s in range(world)]) if world > 1 else rows[0, :sc[0]].contiguous()
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=group)
    return recv


def all_ranks_chunk_count(n_local_chunks, group=None, device="cpu"):
    """every rank must issue the same number of all-to-alls"""
    t = torch.tensor([n_local_chunks], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


MAX_CHUNK_BYTES = 1 << 30      # payload of one all-to-all per rank.  Measured with tools/a2a_check.py (RCCL 2.26.6, one rank): a 1.0 GiB
                               # message arrives intact, at 1.5 GiB half of the rows do not -- so no message may exceed 1 GiB,
                               # although every chunk costs a counts exchange, a host round trip and a launch tail in level 1


def _plan_chunks(offsets_host, n, rb, chunk_reads):
    if chunk_reads is None:
        avg = max(1, int(offsets_host[n] - offsets_host[0]) // max(1, n))
        chunk_reads = max(1024, (MAX_CHUNK_BYTES // rb) // avg)
    n_chunks = (n + chunk_reads - 1) // chunk_reads
    max_kmers = 0
    for c in range(n_chunks):
        lo, hi = c * chunk_reads, min(n, (c + 1) * chunk_reads)
        max_kmers = max(max_kmers, int(offsets_host[hi] - offsets_host[lo]))
    return chunk_reads, n_chunks, max_kmers


def build_partitioned(spectrum, bases, quals, offsets, first_read_idx=0, chunk_reads=None, group=None, slack=1.25, pipeline=True):
    """Device tensors in, spectrum (rank/world_size configured) built in place.
    bases/quals: uint8 cuda tensors, offsets: int64/uint64 cuda tensor [n+1].

    pipeline=True overlaps the all-to-all of chunk c with the extraction of chunk c+1: the library's stream S
    runs extract(0), extract(1), [wait comm 0] insert(0), extract(2), [wait comm 1] insert(1), ... while a second
    stream compacts the owner segments and runs the collectives; two record buffers alternate."""
    from . import record_bytes
    world = dist.get_world_size(group)
    n = offsets.numel() - 1
    rb = record_bytes(spectrum.k, spectrum.cfg.value_kind)
    dev = bases.device
    off_host = offsets.cpu()
    chunk_reads, n_chunks, max_kmers = _plan_chunks(off_host, n, rb, chunk_reads)
    total_chunks = all_ranks_chunk_count(n_chunks, group, dev)
    seg_cap = max(1024, int(max_kmers / world * slack) + 1024)
    nbuf = 2 if pipeline else 1
    records = [torch.empty(world * seg_cap * rb, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    counts = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(nbuf)]
    qptr = None if quals is None else quals.data_ptr()

    def submit_extract(c):
        b = c % nbuf
        lo, hi = c * chunk_reads, min(n, (c + 1) * chunk_reads)
        if lo < n:
            nb = int(off_host[hi] - off_host[lo])
            spectrum.extractByOwnerDevice(bases.data_ptr(), qptr, offsets.data_ptr() + 8 * lo, hi - lo, nb, first_read_idx + lo,
                                          records[b].data_ptr(), seg_cap, counts[b].data_ptr())
        else:
            with torch.cuda.stream(lib_stream):
                counts[b].zero_()

    if not pipeline:
        lib_stream = torch.cuda.current_stream(dev)
        for c in range(total_chunks):
            torch.cuda.synchronize(dev)
            submit_extract(c)
            spectrum.sync()
            recv, n_recv = exchange_records(records[0], counts[0], seg_cap, rb, group)
            if n_recv:
                torch.cuda.synchronize(dev)
                spectrum.insertRecordsDevice(recv.data_ptr(), n_recv)
                spectrum.sync()
        return spectrum

    lib_stream = torch.cuda.ExternalStream(spectrum.stream(), device=dev)
    comm_stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    ev_extract = [torch.cuda.Event() for _ in range(nbuf)]
    keep_alive = []
    words = rb // 4
    submit_extract(0)
    ev_extract[0].record(lib_stream)
    for c in range(total_chunks):
        b = c % nbuf
        if c + 1 < total_chunks:
            submit_extract(c + 1)
            ev_extract[(c + 1) % nbuf].record(lib_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev_extract[b])
            send_counts = counts[b].clone()
            recv_counts = torch.empty_like(send_counts)
            dist.all_to_all_single(recv_counts, send_counts, group=group)
            sc = [int(x) for x in send_counts.cpu().tolist()]          # host waits for extract(c) and the counts exchange only
            rc = [int(x) for x in recv_counts.cpu().tolist()]
            if max(sc) > seg_cap:
                raise RuntimeError("owner segment overflow: %d > %d" % (max(sc), seg_cap))
            rows = records[b].view(torch.int32).view(world, seg_cap, words)
            recv = _all_to_all_rows(rows, sc, rc, group)
            ev_comm = torch.cuda.Event()
            ev_comm.record(comm_stream)
        recv.record_stream(lib_stream)
        keep_alive.append(recv)
        if len(keep_alive) > 3:
            keep_alive.pop(0)
        lib_stream.wait_event(ev_comm)
        if sum(rc):
            spectrum.insertRecordsDevice(recv.data_ptr(), sum(rc))
    spectrum.sync()
    torch.cuda.synchronize(dev)
    return spectrum
