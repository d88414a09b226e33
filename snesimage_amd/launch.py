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


def spawn_ranks(n_ranks, argv, env_extra=None, timeout=None):
    """Run `python argv...` as n_ranks ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as
    torch.distributed.run sets them).  Returns (worst exit code, rank 0's stdout)."""
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs across processes on these hosts
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable, *argv], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, code = b"", 0
    try:
        out, _ = procs[0].communicate(timeout=timeout)
        for p in procs:
            rc = p.wait(timeout=timeout)
            code = rc if abs(rc) > abs(code) else code
    except subprocess.TimeoutExpired:
        code = 124
    finally:
        for p in procs:  # a rank that outlives the job (a failed peer left it waiting) is ended by its own pid, never by pattern
            if p.poll() is None:
                p.kill()
                p.wait()
    return code, out.decode()
