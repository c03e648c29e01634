"""Owner-partitioned spectrum build, one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Replaces the exchange of DistributedKmerSpectrum::_buildKmerSpectrumMPI
(src/DistributedFunctions.h:340-458): every rank extracts the good k-mers of its own
reads, bins them by owner = ((lookup3 >> 24) & 0x7ffff) % world (src/Kmer.h:2284-2295),
and one all-to-all per chunk moves each record to its owner (MPI_Alltoallv of
src/MPIBuffer.h:588-600; the 'int dataSize' prefix becomes the counts all-to-all).
Termination is implicit: every rank runs the same number of chunks.
"""
import numpy as np
import torch
import torch.distributed as dist


def _staged(t, group=None):
    """gloo moves host memory: a device tensor goes through the host there (CPU tests, and rehearsals of the N>1 code with
    several ranks on ONE GPU, which RCCL refuses); RCCL takes device memory as it is"""
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _exchange_counts(counts, group=None):
    """counts: int64 tensor [world], records for each rank -> (sent, received) as Python lists"""
    send = counts.to(torch.int64)
    if _staged(send, group):
        send = send.cpu()
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return [int(x) for x in send.cpu().tolist()], [int(x) for x in recv.cpu().tolist()]


def _check_overflow(sc, seg_capacity, group, device):
    """An owner segment that overflowed has lost records on THAT rank only: the decision to stop is taken by everybody (a MAX
    all-reduce of the flag), otherwise the other ranks would walk into the records all-to-all and wait for the one that raised."""
    flag = torch.tensor([1 if max(sc) > seg_capacity else 0], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()):
        raise RuntimeError("owner segment overflow on some rank (this rank: %d of %d records): raise `slack` or lower `chunk_reads`" % (max(sc), seg_capacity))


def exchange_records(records, seg_counts, seg_capacity, rec_bytes, group=None):
    """records: uint8 tensor [world * seg_capacity * rec_bytes], segment s holds
    seg_counts[s] records for rank s.  Returns (recv uint8 tensor, n_records).

    The payload travels as int32 rows of one record each, so the split sizes handed to the
    collective count records (a byte count overflows 32 bits at ~1.8e8 twelve-byte records)."""
    world = dist.get_world_size(group)
    sc, rc = _exchange_counts(seg_counts, group)
    _check_overflow(sc, seg_capacity, group, records.device)
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
        if _staged(send, group):
            got = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(got, send.cpu(), output_split_sizes=rc, input_split_sizes=sc, group=group)
            recv.copy_(got)
        else:
            dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=group)
    return recv


def all_ranks_chunk_count(n_local_chunks, group=None, device="cpu"):
    """every rank must issue the same number of all-to-alls"""
    t = torch.tensor([n_local_chunks], dtype=torch.int64, device=device if dist.get_backend(group) == "nccl" else "cpu")
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


def build_partitioned(spectrum, bases, quals, offsets, first_read_idx=0, chunk_reads=None, group=None, slack=1.25, pipeline=True, stats=None):
    """Device tensors in, spectrum (rank/world_size configured) built in place.
    bases/quals: uint8 cuda tensors, offsets: int64/uint64 cuda tensor [n+1].

    pipeline=True overlaps the all-to-all of chunk c with the extraction of chunk c+1: the library's stream S
    runs extract(0), extract(1), [wait comm 0] insert(0), extract(2), [wait comm 1] insert(1), ... while a second
    stream compacts the owner segments and runs the collectives; two record buffers alternate.

    stats (a dict, optional) accumulates over calls: "chunks", "records_sent", "bytes_to_peers" (what leaves this GPU over xGMI:
    the records of other owners), "alltoall_ms" (device time of the record all-to-alls, by events on their stream)."""
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
    timed = []
    my_rank = dist.get_rank(group)
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
            sc, rc = _exchange_counts(counts[b].clone(), group)     # host waits for extract(c) and the counts exchange only
            _check_overflow(sc, seg_cap, group, dev)
            rows = records[b].view(torch.int32).view(world, seg_cap, words)
            ev_a = torch.cuda.Event(enable_timing=True)
            ev_a.record(comm_stream)
            recv = _all_to_all_rows(rows, sc, rc, group)
            ev_comm = torch.cuda.Event(enable_timing=True)
            ev_comm.record(comm_stream)
            timed.append((ev_a, ev_comm))
            if stats is not None:
                stats["chunks"] = stats.get("chunks", 0) + 1
                stats["records_sent"] = stats.get("records_sent", 0) + sum(sc)
                stats["bytes_to_peers"] = stats.get("bytes_to_peers", 0) + (sum(sc) - sc[my_rank]) * rb
        recv.record_stream(lib_stream)
        keep_alive.append(recv)
        if len(keep_alive) > 3:
            keep_alive.pop(0)
        lib_stream.wait_event(ev_comm)
        if sum(rc):
            spectrum.insertRecordsDevice(recv.data_ptr(), sum(rc))
    spectrum.sync()
    torch.cuda.synchronize(dev)
    if stats is not None:
        stats["alltoall_ms"] = stats.get("alltoall_ms", 0.0) + sum(a.elapsed_time(b) for a, b in timed)
    return spectrum


def build_partitioned_superkmers(spectrum, bases, quals, offsets, first_read_idx=0, group=None, stats=None, stream_origin=None, pieces=1,
                                 list_steps=1, early_min_depth=None):
    """The N > 1 build on super-k-mer lists (kmr_config.build_mode = 3, rank / world_size configured): every rank scatters the
    super-k-mers of its own reads into the job's lists on its own GPU (no owner filter), list l belongs to rank l % world, and
    the chunks a rank holds of other ranks' lists travel as they lie -- one all-to-all of (list, granules) pairs and one of the
    16-byte granules, ~4 bytes per k-mer instead of the 12-byte k-mer records of build_partitioned (24 + kb in the reference,
    src/DistributedFunctions.h:274-303).  The owner appends what it receives to its own lists; kmr_finalize then counts them as
    in the single-GPU build.  Replaces _buildKmerSpectrumMPI (src/DistributedFunctions.h:340-458).

    bases / quals: uint8 tensors on the spectrum's device, offsets: int64 tensor [n + 1].  stream_origin: position of this rank's
    first base in the whole input (default: the bases of the lower ranks, found by an all-gather) -- with it the first sighting of
    a k-mer is the first one in the whole input, as in a serial build, whatever the ranks' timing.  pieces > 1 (the same on every
    rank): the reads go through in that many pieces and the all-to-all of piece i runs (RCCL's own stream) while the library's
    stream extracts piece i + 1 -- over xGMI the exchange of a C2 batch takes about as long as its extraction.  stats (a dict,
    optional) accumulates "bytes_to_peers", "records_sent" (granules), "chunks", "alltoall_ms".

    list_steps > 1 (with pieces == 1; the same on every rank): the whole batch is extracted first and the exchange then goes over the
    list space in that many steps (kmr_sk_exchange_range) -- while step s + 1 is on the wire the owner has everything the lists of
    steps <= s will ever get and, given early_min_depth (the min_depth kmr_finalize will be called with), counts them
    (kmr_count_lists_prefix): the count pass of the lower part of the list space overlaps the all-to-all of the upper part.  The
    result is that of the one-step exchange.  Not measured on real devices: off by default."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = bases.device
    n = offsets.numel() - 1
    nccl = dist.get_backend(group) == "nccl"
    pieces = max(1, int(pieces))
    cuts = [min(n, ((n * i // pieces) + 63) // 64 * 64) for i in range(pieces)] + [n]
    cut_off = [int(x) for x in offsets[torch.tensor(cuts, dtype=torch.int64, device=offsets.device)].cpu().tolist()] if n else [0] * (pieces + 1)
    total = cut_off[-1] - cut_off[0]
    if stream_origin is None:
        mine = torch.tensor([total], dtype=torch.int64, device=dev if nccl else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine, group=group)
        stream_origin = sum(int(t.item()) for t in every[:rank])
    spectrum.sk_exchange_begin()

    def start(i):
        """close piece i's lists, pack what belongs to others, exchange the counts, set the two all-to-alls going"""
        chunks, granules = spectrum.sk_exchange_counts()
        send_c = [int(chunks[r]) if r != rank else 0 for r in range(world)]
        send_g = [int(granules[r]) if r != rank else 0 for r in range(world)]
        goff, coff, ag, ac = [], [], 0, 0
        for r in range(world):
            goff.append(ag)
            coff.append(ac)
            ag += send_g[r]
            ac += send_c[r]
        data = torch.empty((max(ag, 1), 4), dtype=torch.int32, device=dev)
        meta = torch.empty((max(ac, 1), 2), dtype=torch.int32, device=dev)
        spectrum.sk_exchange_pack(data.data_ptr(), meta.data_ptr(), goff, coff)
        # (a sender's uniform-weight state rides in the upper bits of its chunk counts -- chunk counts stay below 2^24 per owner and
        # batch: the owner can then take the count pass's one-weight form without looking at what it receives)
        assert max(send_c + [0]) < (1 << 24), "more than 2^24 chunks for one owner in one piece: feed smaller pieces"
        ust = spectrum.sk_exchange_uniform()
        _, recv_enc = _exchange_counts(torch.tensor([c | (ust << 24) for c in send_c], dtype=torch.int64, device=dev), group)
        recv_c = [x & 0xFFFFFF for x in recv_enc]
        peer_states = [x >> 24 for x in recv_enc]
        _, recv_g = _exchange_counts(torch.tensor(send_g, dtype=torch.int64, device=dev), group)
        timed = None
        if dev.type == "cuda":
            timed = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            timed[0].record()
        works = []
        overlap = pieces > 1 or list_steps > 1
        got_meta = _all_to_all_sliced(meta[:ac], send_c, recv_c, group, works=works if overlap else None)
        got_data = _all_to_all_sliced(data[:ag], send_g, recv_g, group, works=works if overlap else None)
        if stats is not None:
            stats["chunks"] = stats.get("chunks", 0) + sum(send_c)
            stats["records_sent"] = stats.get("records_sent", 0) + sum(send_g)
            stats["bytes_to_peers"] = stats.get("bytes_to_peers", 0) + 16 * sum(send_g) + 8 * sum(send_c)
        return dict(meta=got_meta, data=got_data, recv_c=recv_c, recv_g=recv_g, works=works, timed=timed, keep=(data, meta), peer_states=peer_states)

    def finish(x):
        """wait for the piece's all-to-alls, append what arrived to this rank's lists"""
        for w in x["works"]:
            w[0].wait()
        if x["timed"]:
            x["timed"][1].record()
            torch.cuda.synchronize(dev)
            if stats is not None:
                stats["alltoall_ms"] = stats.get("alltoall_ms", 0.0) + x["timed"][0].elapsed_time(x["timed"][1])
        for r, n_c in enumerate(x["recv_c"]):
            if r != rank and n_c:
                spectrum.sk_exchange_peer_uniform(x["peer_states"][r])
        if sum(x["recv_c"]):
            spectrum.sk_exchange_adopt(x["data"].data_ptr(), x["meta"].data_ptr(), sum(x["recv_c"]), sum(x["recv_g"]))

    list_steps = max(1, int(list_steps))
    if list_steps > 1:
        if pieces != 1:
            raise ValueError("list_steps > 1 goes with pieces == 1")
        if n:
            spectrum.set_stream_origin(stream_origin)
            spectrum.buildKmerSpectrumDevice(bases.data_ptr(), None if quals is None else quals.data_ptr(), offsets.data_ptr(), n, total, first_read_idx)
        nl = int(spectrum.build_info("lists"))
        bounds = [nl * s // list_steps for s in range(list_steps)] + [nl]
        pending, pending_hi = None, 0
        for s in range(list_steps):
            spectrum.sk_exchange_range(bounds[s], bounds[s + 1] if s + 1 < list_steps else 0xFFFFFFFFFFFFFFFF)
            x = start(s)                       # counts, pack, the step's all-to-alls set going (behind the step before on the wire)
            if pending is not None:
                finish(pending)                # the step before has arrived: adopt it ...
                if early_min_depth is not None:
                    spectrum.count_lists_prefix(early_min_depth, pending_hi)      # ... and count everything below its bound while this step travels
            pending, pending_hi = x, bounds[s + 1]
        finish(pending)
        spectrum.sk_exchange_range()
        return spectrum
    pending = None
    for i in range(pieces):
        lo, hi = cuts[i], cuts[i + 1]
        if hi > lo:
            # the handle adds a call's bases to its stream position; the offsets handed in are absolute, so the origin stays put
            spectrum.set_stream_origin(stream_origin)
            spectrum.buildKmerSpectrumDevice(bases.data_ptr(), None if quals is None else quals.data_ptr(), offsets.data_ptr() + 8 * lo, hi - lo,
                                             cut_off[i + 1] - cut_off[i], first_read_idx + lo)
        if pending is not None:
            finish(pending)
        pending = start(i)
    finish(pending)
    return spectrum


def _all_to_all_sliced(send, send_split, recv_split, group=None, max_rows=None, works=None):
    """_all_to_all_flat in slices, so that no single message exceeds MAX_CHUNK_BYTES (see there); every rank runs the same number of
    slices.  send: [sum(send_split), words] rows grouped by destination; returns the received rows grouped by source.  works (a
    list, RCCL only): the collectives are started with async_op and (work, tensors to keep alive) pairs are appended to it -- the
    returned tensor is complete once every work has been waited for (a single slice only; more slices run one after the other)."""
    world = len(send_split)
    words = send.shape[1] if send.dim() > 1 else 1
    if max_rows is None:
        max_rows = max(1, MAX_CHUNK_BYTES // (4 * words))
    need = max([0] + [(x + max_rows - 1) // max_rows for x in list(send_split) + list(recv_split)])
    t = torch.tensor([need], dtype=torch.int64, device=send.device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    slices = max(1, int(t.item()))

    def one(piece, ss, rs):      # rows are contiguous and grouped by destination: the single-tensor collective with split sizes
        got = torch.empty((sum(rs), words), dtype=piece.dtype, device=piece.device)
        if _staged(piece, group):
            host = torch.empty(got.shape, dtype=got.dtype)
            dist.all_to_all_single(host, piece.cpu().contiguous(), output_split_sizes=rs, input_split_sizes=ss, group=group)
            got.copy_(host)
        elif works is not None and slices == 1:
            src = piece.contiguous()
            works.append((dist.all_to_all_single(got, src, output_split_sizes=rs, input_split_sizes=ss, group=group, async_op=True), (src, got)))
        else:
            dist.all_to_all_single(got, piece.contiguous(), output_split_sizes=rs, input_split_sizes=ss, group=group)
        return got

    if world == 1:
        return send[:0].reshape(0, words)
    if slices == 1:
        return one(send, list(send_split), list(recv_split))
    out = torch.empty((sum(recv_split), words), dtype=send.dtype, device=send.device)
    sbase, rbase, a, b = [], [], 0, 0
    for r in range(world):
        sbase.append(a)
        rbase.append(b)
        a += send_split[r]
        b += recv_split[r]
    for p in range(slices):
        ss = [max(0, min(max_rows, send_split[r] - p * max_rows)) for r in range(world)]
        rs = [max(0, min(max_rows, recv_split[r] - p * max_rows)) for r in range(world)]
        piece = torch.cat([send[sbase[r] + p * max_rows: sbase[r] + p * max_rows + ss[r]] for r in range(world)]) if sum(ss) else send[:0]
        got = one(piece, ss, rs)
        at = 0
        for r in range(world):
            if rs[r]:
                out[rbase[r] + p * max_rows: rbase[r] + p * max_rows + rs[r]] = got[at:at + rs[r]]
            at += rs[r]
    return out


def _all_to_all_flat(send, send_split, recv_split, group=None):
    """send: [sum(send_split), words] rows grouped by destination; returns [sum(recv_split), words] grouped by source"""
    recv = torch.empty((sum(recv_split), send.shape[1]), dtype=send.dtype, device=send.device)
    if dist.get_backend(group) == "nccl":
        dist.all_to_all(list(recv.split(recv_split)), list(send.split(send_split)), group=group)
    elif _staged(send, group):
        got = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(got, send.cpu().contiguous(), output_split_sizes=recv_split, input_split_sizes=send_split, group=group)
        recv.copy_(got)
    else:
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_split, input_split_sizes=send_split, group=group)
    return recv


def score_partitioned(spectrum, bases, offsets, minimum_kmer_score, scoring_type="MEDIAN", chunk_reads=None, group=None, slack=1.25):
    """scoreAndTrimReads over an owner-partitioned spectrum: DistributedReadSelector::scoreAndTrimReads
    (src/DistributedFunctions.h:900-1045).  Every rank scores its own reads; a k-mer is looked up on the rank that owns it.
    Per chunk of reads: the k-mers binned by owner (device) -> all-to-all of the keys (8 * words bytes each; the reference sends
    requestId + k-mer) -> weak-map lookup at the owner (device) -> all-to-all of the u32 counts back, in request order -> scatter
    to the k-mers' positions (device); then the usual trim + score kernel over the position-indexed counts.

    `spectrum`: finalized, rank / world_size configured (anything with lookup_requests / lookup_keys / scatter_counts /
    score_counts / sync and .k, as KmerSpectrum has).  bases: uint8 tensor, offsets: int64/uint64 tensor [n+1] starting at 0,
    both on the spectrum's device.  Returns (trim_offset, trim_length, score, was_trimmed) host arrays for this rank's reads."""
    world = dist.get_world_size(group)
    dev = bases.device
    n = offsets.numel() - 1
    words = (((spectrum.k + 3) // 4) + 7) // 8
    off_host = offsets.cpu().to(torch.int64)
    total = int(off_host[n]) if n else 0
    if total >= 1 << 32:
        raise RuntimeError("score_partitioned: positions are 32-bit, pass at most 2^32 - 1 bases per call")
    chunk_reads, n_chunks, max_kmers = _plan_chunks(off_host, n, 8 * words, chunk_reads) if n else (1, 0, 0)
    total_chunks = all_ranks_chunk_count(n_chunks, group, dev)
    seg_cap = max(1024, int(max_kmers / world * slack) + 1024)
    keys = torch.empty((world, seg_cap, words), dtype=torch.int64, device=dev)
    pos = torch.empty((world, seg_cap), dtype=torch.int32, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    position_counts = torch.zeros(max(total, 1), dtype=torch.int32, device=dev)

    def fence():
        spectrum.sync()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    fence()
    for c in range(total_chunks):
        lo, hi = c * chunk_reads, min(n, (c + 1) * chunk_reads)
        if lo < n:
            spectrum.lookup_requests(bases, offsets, lo, hi, int(off_host[hi] - off_host[lo]), keys, pos, seg_cap, counts)
        else:
            counts.zero_()
        fence()
        sc, rc = _exchange_counts(counts, group)
        _check_overflow(sc, seg_cap, group, dev)
        asked = _all_to_all_rows(keys, sc, rc, group)                      # the k-mers other ranks want from this one
        answers = torch.zeros((sum(rc), 1), dtype=torch.int32, device=dev)
        if sum(rc):
            fence()
            spectrum.lookup_keys(asked, sum(rc), answers)
            fence()
        back = _all_to_all_flat(answers, rc, sc, group)                    # the counts of this rank's requests, owner by owner
        fence()
        at = 0
        for s in range(world):
            if sc[s]:
                spectrum.scatter_counts(back[at:at + sc[s]], pos[s], sc[s], position_counts)
            at += sc[s]
        fence()
    return spectrum.score_counts(bases, offsets, n, position_counts, minimum_kmer_score, scoring_type)


def reduce_stats(spectrum, group=None):
    """Job-wide counters of an owner-partitioned spectrum: the sums DistributedKmerSpectrum forms over its ranks (raw / rawGood /
    unique / singleton k-mers and map sizes; src/DistributedFunctions.h:460-512 reduces them inside its SizeTracker, the purge
    bookkeeping at :707-710).  Every rank gets the same dict."""
    st = spectrum.stats()
    keys = sorted(st)
    t = torch.tensor([st[k] for k in keys], dtype=torch.int64)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return {k: int(v) for k, v in zip(keys, t.cpu().tolist())}


def reduce_histogram(spectrum, zoom_max=255, log_base=2.0, group=None):
    """MPIHistogram::reduce (src/DistributedFunctions.h:513-535) over kmr_histogram: the three arrays of every rank's
    KmerSpectrum::Histogram (visits, visitedCount, visitedWeight) summed over the ranks -- three all-reduces of ~65.8 k entries,
    as in the reference (DistributedKmerSpectrum::_getHistogram uses Histogram(255), :574-583).  Returns the same Histogram
    object on every rank; .toString() prints the reference's table."""
    from .spectrum import Histogram
    h = spectrum.getHistogram(zoom_max, log_base)
    on_dev = dist.get_backend(group) == "nccl"
    out = []
    for arr, dt in ((h.visits, torch.int64), (h.visitedCount, torch.int64), (h.visitedWeight, torch.float64)):
        t = torch.from_numpy(arr.astype("int64" if dt == torch.int64 else "float64"))
        if on_dev:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        out.append(t.cpu().numpy())
    return Histogram(zoom_max, log_base, out[0].astype("uint64"), out[1].astype("uint64"), out[2])


def reduce_size_tracker(tracker, group=None):
    """DistributedKmerSpectrum::reduceSizeTracker (src/DistributedFunctions.h:460-491): the ranks' size histories summed element by
    element; a rank with fewer elements than the longest repeats its last one.  `tracker`: this rank's kmernator_amd.spectrum
    SizeTracker (a rank's own history comes from a single-partition handle: kmr_create refuses size_tracker with world_size > 1).
    Every rank gets the same SizeTracker."""
    from .spectrum import SizeTracker
    nccl = dist.get_backend(group) == "nccl"
    el = torch.from_numpy(np.ascontiguousarray(tracker.elements, dtype=np.uint64).astype(np.int64))
    n = torch.tensor([el.shape[0]], dtype=torch.int64)
    if nccl:
        n = n.cuda()
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    ct = int(n.item())
    padded = torch.zeros((ct, 4), dtype=torch.int64)
    if el.shape[0]:
        padded[:el.shape[0]] = el
        padded[el.shape[0]:] = el[-1]
    if nccl:
        padded = padded.cuda()
    dist.all_reduce(padded, op=dist.ReduceOp.SUM, group=group)
    return SizeTracker(padded.cpu().numpy().astype(np.uint64))
