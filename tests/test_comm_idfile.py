"""The ncclUniqueId hand-over of bsfm_comm_create_from_env (csrc/idfile.h) between two PROCESSES, without RCCL or a GPU
("stale file, late rank, missing rank => time-out message, symlink refused").  The production caller is
csrc/comm.hip; these tests drive the same functions through bsfm_comm_idfile_exchange."""
import os
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = 0x6273666d5f6e6363

CHILD = r"""
import ctypes as C, os, sys, time
sys.path.insert(0, {root!r})
import bundler_sfm_amd._lib as L
path, rank, world, tmo, grace, delay = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6])
time.sleep(delay)
buf = (C.c_ubyte * 128)(*([(7 * i + 3) % 251 for i in range(128)] if rank == 0 else [0] * 128))
rc = L.lib.bsfm_comm_idfile_exchange(path.encode(), rank, world, tmo, grace, buf)
print("RC", rc, bytes(buf).hex())
"""


def spawn(path, rank, world=2, tmo=5.0, grace=120.0, delay=0.0):
    return subprocess.Popen([sys.executable, "-c", CHILD.format(root=ROOT), path, str(rank), str(world), str(tmo), str(grace), str(delay)],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def result(p, timeout=60):
    out, err = p.communicate(timeout=timeout)
    line = [l for l in out.splitlines() if l.startswith("RC ")][-1].split()
    return int(line[1]), bytes.fromhex(line[2]), err


EXPECTED = bytes((7 * i + 3) % 251 for i in range(128))


def record(world, created_ns, payload=b"\x55" * 128):
    return struct.pack("<QiiQ", MAGIC, world, 0, created_ns) + payload


def write_record(path, data, mode=0o600):
    with open(os.open(path, os.O_WRONLY | os.O_CREAT, mode), "wb") as f:
        f.write(data)
    os.chmod(path, mode)


def test_two_processes_exchange_the_id(tmp_path):
    path = str(tmp_path / "job.id")
    r1 = spawn(path, 1)                  # the waiting rank starts first
    r0 = spawn(path, 0, delay=0.5)       # rank 0 is late
    rc0, _, _ = result(r0)
    rc1, got, err = result(r1)
    assert rc0 == 0 and rc1 == 0, err
    assert got == EXPECTED
    st = os.stat(path)
    assert (st.st_mode & 0o777) == 0o600 and st.st_uid == os.geteuid()


def test_a_stale_id_of_an_earlier_job_is_ignored_until_rank0_replaces_it(tmp_path):
    path = str(tmp_path / "job.id")
    write_record(path, record(2, time.time_ns() - 3600 * 10**9))    # an hour old: a crashed job on the same address / port
    r1 = spawn(path, 1, tmo=8.0, grace=5.0)
    time.sleep(1.0)
    r0 = spawn(path, 0)
    assert result(r0)[0] == 0
    rc1, got, err = result(r1)
    assert rc1 == 0 and got == EXPECTED, err                          # not the stale 0x55 payload


def test_a_missing_rank0_ends_in_a_timeout_that_names_the_file(tmp_path):
    path = str(tmp_path / "nobody.id")
    t0 = time.time()
    rc, _, err = result(spawn(path, 1, tmo=1.5))
    assert rc != 0 and 1.0 < time.time() - t0 < 30.0
    assert "found no fresh id" in err and path in err


def test_symlinks_foreign_world_sizes_and_open_permissions_are_refused(tmp_path):
    good = record(2, time.time_ns())
    # (a) a symbolic link to an otherwise perfect record
    target = str(tmp_path / "target.id")
    write_record(target, good)
    link = str(tmp_path / "link.id")
    os.symlink(target, link)
    rc, _, err = result(spawn(link, 1, tmo=1.0))
    assert rc != 0 and "symbolic link" in err
    # (b) a record of a job with another world size
    rc, _, err = result(spawn(target, 1, world=4, tmo=1.0))
    assert rc != 0 and "another job" in err
    # (c) group / other permission bits
    loose = str(tmp_path / "loose.id")
    write_record(loose, good, 0o644)
    rc, _, err = result(spawn(loose, 1, tmo=1.0))
    assert rc != 0 and "others" in err
    # (d) the same record with mode 0600 is accepted
    rc, got, err = result(spawn(target, 1, tmo=2.0))
    assert rc == 0 and got == b"\x55" * 128, err


def test_rank0_replaces_a_symlink_planted_at_its_path(tmp_path):
    victim = str(tmp_path / "victim.txt")
    with open(victim, "w") as f:
        f.write("do not touch")
    path = str(tmp_path / "job.id")
    os.symlink(victim, path)
    assert result(spawn(path, 0))[0] == 0
    assert not os.path.islink(path) and open(victim).read() == "do not touch"
