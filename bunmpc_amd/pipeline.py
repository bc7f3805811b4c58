"""Several batches in flight on one GPU.  A batched `KinoDynMP.optimize` ends in a tail: a few problems still iterating
their DDP while most of the chip idles (the per-iteration latency of a lone problem sets the pace).  Independent batches
on separate HIP streams fill that idle time -- the tail of one overlaps the bulk phases of the others.  Each batch gets
a host thread because the batched DDP loop is host-driven (it looks at the active-problem counter between iterations):
the C-ABI call releases the GIL and waits on its own stream only, its scratch state is thread-local.

Measured (MI355X, `bench.py` `multi_stream`): 3 streams give 1.2-1.3x the single-stream throughput on Solo12 H=20 / H_ik=10
(B = 4096 per batch) and 1.7x on the synthetic Go2 H=60 / H_ik=30 (B = 1024); a 4th stream loses again (HIP multiplexes
streams onto 4 hardware queues, one of which the default stream holds)."""
import threading

DEFAULT_STREAMS = 3
# The batched DDP gives every problem a second wave for the Riccati gains once few problems are left: that shortens ONE
# batch's tail by using SIMDs that idle -- with several batches in flight they do not idle, the other batches use them
# (measured, Go2 H = 60, three batches: 2.02e4 solves/s without, 1.97e4 at a third of the threshold, 1.90e4 with).
# ... and the express lane (a persistent kernel on a side stream per batch) is for a batch that has the chip to itself
IN_FLIGHT_SCHEDULE = {"gains_wave_below": -1, "express_cap": -1}


class StreamPool:
    def __init__(self, device="cuda:0", n_streams=DEFAULT_STREAMS):
        import torch
        self.torch, self.device = torch, torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]

    def run(self, jobs):
        """jobs: callables taking no argument (e.g. `batch.solve`); job i runs on stream i mod n, jobs that share a
        stream one after the other.  Returns once everything has finished on the device; re-raises the first error."""
        torch = self.torch
        lanes = [[] for _ in self.streams]
        for i, j in enumerate(jobs):
            lanes[i % len(self.streams)].append(j)
        errors = []

        def worker(stream, todo):
            try:
                with torch.cuda.stream(stream):
                    for j in todo:
                        j()
                stream.synchronize()
            except BaseException as e:       # noqa: BLE001 -- handed to the caller
                errors.append(e)
        threads = [threading.Thread(target=worker, args=(s, l)) for s, l in zip(self.streams, lanes) if l]
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)               # inputs prepared on the caller's stream are visible to the jobs
        # (Scheduling of the jobs' DDP loops is the jobs' own: batches meant to run several at a time are created with
        # schedule=IN_FLIGHT_SCHEDULE -- a per-batch field of bmpc_ik_batch_t, not a process-wide switch flipped around the run.)
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for s in self.streams:
            cur.wait_stream(s)
        if errors:
            raise errors[0]
