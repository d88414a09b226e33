"""One process per GPU without an external launcher.

`python bench.py --gpus N` must work as typed.  When the rendezvous environment of `torch.distributed.run` is absent
(WORLD_SIZE unset) the parent process starts N fresh children — one rank per GPU, rendezvous on 127.0.0.1 — BEFORE it has
touched the GPU itself (a process that has initialised HIP must never be replaced or forked into ranks), relays the ranks'
stderr, keeps rank 0's stdout as the job's stdout (the one JSON line) and exits with the worst exit code.
This module imports nothing that initialises a device.
"""
import os
import socket
import subprocess
import sys


def needs_spawn(n_ranks, environ=None):
    """True when this process was started plainly (no torchrun environment) but is asked for several ranks."""
    env = os.environ if environ is None else environ
    return n_ranks > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n_ranks, argv, env_extra=None, timeout=None, poll_s=0.05):
    """Run `python argv...` as n_ranks ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as
    torch.distributed.run sets them).  Returns (exit code, rank 0's stdout).

    Every rank is watched, not rank 0 alone: a rank that dies leaves its peers waiting in the rendezvous or in a collective
    for ever, so the first rank that exits non-zero ends the job — the others are terminated by pid, one line on stderr says
    which rank failed with what, and its code is the job's.  `timeout` (seconds, None = unlimited) bounds the whole job: 124."""
    import tempfile
    import time
    port = free_port()
    procs, out0 = [], tempfile.TemporaryFile()  # rank 0's stdout goes to a file: nobody has to drain a pipe while the ranks are watched
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs across processes on these hosts
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable, *argv], env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
    code, t0 = 0, time.monotonic()
    try:
        live = set(range(n_ranks))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc
                    if live:
                        sys.stderr.write("launch: rank %d of %d exited with code %d; terminating ranks %s\n" % (r, n_ranks, rc, sorted(live)))
                        sys.stderr.flush()
            if code != 0:
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                sys.stderr.write("launch: job still running after %.0f s; terminating ranks %s\n" % (timeout, sorted(live)))
                sys.stderr.flush()
                code = 124
                break
            if live:
                time.sleep(poll_s)
    finally:
        for p in procs:  # a rank that outlives the job (a failed peer left it waiting) is ended by its own pid, never by pattern
            if p.poll() is None:
                p.terminate()
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
    out0.seek(0)
    out = out0.read()
    out0.close()
    return code, out.decode()
