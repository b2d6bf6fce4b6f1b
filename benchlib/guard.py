"""The ONE JSON line must reach stdout whatever happens to a secondary object: emergency writer, store-based verdicts, test hooks."""
import json
import os


class _Emergency:
    """The headline must reach stdout whatever happens to a secondary object.  At N > 1 a rank that
    dies inside a sharded object leaves the others inside a collective; the launcher then SIGTERMs them.  Rank 0 therefore
    arms, as soon as the headline is computed, a watcher THREAD on the signal wake-up pipe (a Python-level handler would
    not run while the main thread sits in a collective / device synchronisation; those calls release the GIL, so a thread
    does): on SIGTERM / SIGINT it writes the ONE JSON line -- headline + {"error": ...} for the object in flight -- and exits."""

    def __init__(self, json_fd, render=None, on_emit=None):
        """`render(out) -> str` turns the collected objects into the line (default: json.dumps of everything); `on_emit(out)` runs
        first (bench.py files the full objects away there) and may fail without costing the line."""
        import threading
        self.json_fd, self.out, self.stage, self.done = json_fd, None, "headline", False
        self.render = render or (lambda out: json.dumps(out, default=repr))
        self.on_emit = on_emit
        self.lock = threading.Lock()
        self.fallback = None

    def arm(self, out):
        import signal
        import threading
        self.out = out
        # what is printed if nothing else can be: the headline as it stands now (the main thread keeps adding objects to `out`)
        self.fallback = self.render(dict(out))
        r, w = os.pipe()
        os.set_blocking(w, False)
        signal.set_wakeup_fd(w, warn_on_full_buffer=False)
        for sig in (signal.SIGTERM, signal.SIGINT):
            signal.signal(sig, lambda *a: None)          # the C-level handler writes the signal number to the pipe

        def watch():
            os.read(r, 1)
            self.emit(f"the job was terminated during '{self.stage}' (a rank left it; signal from the launcher)", code=3)

        threading.Thread(target=watch, daemon=True).start()

    def emit(self, error=None, code=None):
        """Write the line exactly once and, with `code`, leave the process -- whatever happens on the way: the watcher thread
        may run this while the main thread is still adding objects to `out` (a dict that changes size under json.dumps raises),
        so a shallow copy is serialised, retried, and the headline kept at arm() time is the last resort."""
        try:
            with self.lock:
                if self.done or self.out is None:
                    return
                self.done = True
            text = None
            for _ in range(5):
                try:
                    snap = dict(self.out)
                    if error is not None:
                        snap["errors"] = list(snap.get("errors", [])) + [error]
                        if self.stage not in snap:
                            snap[self.stage] = {"error": error}
                    _annotate_ports(snap)
                    if self.on_emit is not None:
                        try:
                            self.on_emit(snap)
                        except Exception:  # noqa: BLE001 -- filing the full objects away is optional
                            pass
                    text = self.render(snap)
                    break
                except Exception:  # noqa: BLE001 -- a concurrent mutation, an unserialisable object: try again, then fall back
                    text = None
            if text is None:
                text = self.fallback if self.fallback is not None else json.dumps({"error": error or "the line could not be serialised"})
            os.write(self.json_fd, (text + "\n").encode())
        finally:
            if code is not None:
                os._exit(code)


PORT_OVER_REFERENCE_TIME = 1.15   # profiles/r04_port_vs_reference_cpu.json: the oracle (kind "port") takes 1.14 - 1.16 x the time of the
                                  # unmodified reference modules on the same host (bit-identical outputs; first-touch of its activations)


def _annotate_ports(node):
    """Every CPU leg of kind "port" says by how much the port understates the reference's own CPU rate, so that no
    GPU / CPU ratio on the line is read as more than an upper bound."""
    if isinstance(node, dict):
        if node.get("kind") == "port" and "port_over_reference_time" not in node:
            node["port_over_reference_time"] = PORT_OVER_REFERENCE_TIME
            node["port_note"] = ("the port runs 1.14-1.16x the unmodified reference's time on the same host (profiles/r04_port_vs_reference_cpu.json): "
                                 "ratios against this leg are upper bounds by that factor")
        for v in list(node.values()):
            _annotate_ports(v)
    elif isinstance(node, list):
        for v in node:
            _annotate_ports(v)


def _inject(name, rank):
    """Test hook (tests/test_gpu_dist.py): NM_BENCH_INJECT_FAILURE="mesh:1" raises inside that object on that rank,
    "buff:all" on every rank."""
    spec = os.environ.get("NM_BENCH_INJECT_FAILURE", "")
    for item in spec.split(","):
        obj, _, who = item.partition(":")
        if obj == name and who in ("all", str(rank)):
            raise RuntimeError(f"injected failure in '{name}' on rank {rank}")


def _guarded(name, fn, rank, world, emergency, healthy_wait_s=1800, failed_wait_s=45):
    """Run one secondary object so that its failure cannot take the line down or hang the job.  The object's own
    collectives run on RCCL; the VERDICT on the object travels through the rendezvous store (no collective a failed rank could
    mismatch): every rank posts "" or its error after leaving the object and waits for the others' posts.
      * all ranks fail at the same place (a bug, an out-of-memory at this size): all post promptly, all skip together,
        the line carries {"error": ...} for the object and the next object runs;
      * one rank fails while the others sit in a collective it never joins: its wait for their posts times out
        (`failed_wait_s`), it leaves the job, the launcher terminates the rest and rank 0's emergency writer emits the line."""
    import datetime
    emergency.stage = name
    err, res = None, None
    try:
        _inject(name, rank)
        res = fn()
    except Exception as e:  # noqa: BLE001 -- the headline line must not depend on a secondary figure
        err = repr(e)
    if world == 1:
        return {"error": err} if err else res
    from torch.distributed.distributed_c10d import _get_default_store
    store = _get_default_store()
    keys = [f"nm_bench/{name}/{r}" for r in range(world)]
    store.set(keys[rank], err or "")
    try:
        store.wait(keys, datetime.timedelta(seconds=failed_wait_s if err else healthy_wait_s))
    except Exception:  # noqa: BLE001 -- the others never left the object: they are inside a collective this rank abandoned
        msg = f"rank {rank} failed in '{name}' ({err}) while other ranks were inside a collective" if err else \
              f"rank {rank}: other ranks never left '{name}'"
        if rank == 0:
            emergency.emit(msg, code=3)
        os._exit(3)
    errs = {r: store.get(k).decode() for r, k in enumerate(keys)}
    failed = {r: e for r, e in errs.items() if e}
    if failed:
        return {"error": next(iter(failed.values())), "failed_ranks": sorted(failed)}
    return res
