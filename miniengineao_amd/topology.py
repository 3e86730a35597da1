"""Host placement of a rank: bind the process that drives GPU `device` to the CPUs of the NUMA node the GPU hangs off.

The eight GPUs of an MI355X node hang off two sockets; a rank whose host thread enqueues launches and waits on events from the
other socket pays a cross-socket hop on every doorbell and every event read.  The node of a device comes from the C ABI
(meao_device_numa_node: sysfs numa_node of the device's PCI function), so this host and the in-process pool (meao_pool_*,
whose worker threads bind themselves the same way) agree.  Nothing is bound where the kernel knows no node (-1: single-node
hosts, most VMs) or where the allowed CPU mask (cgroup / taskset) has no CPU on the node."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Set

from . import _lib as L


def parse_cpulist(text: str) -> Set[int]:
    """The kernel's cpulist form, "0-15,32-47" -> {0..15, 32..47}."""
    cpus: Set[int] = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_placement(device: int) -> dict:
    """{"device", "numa_node", "cpulist"} of a visible HIP device (numa_node -1 = unknown); needs the HIP runtime."""
    lib = L.load()
    node = C.c_int32(-1)
    buf = C.create_string_buffer(4096)
    status = lib.meao_device_numa_node(device, C.byref(node), buf, len(buf))
    if status != L.OK:
        return {"device": device, "numa_node": None, "cpulist": "", "status": status}
    return {"device": device, "numa_node": node.value, "cpulist": buf.value.decode()}


def plan_binding(placement: dict, allowed: Optional[Set[int]] = None) -> dict:
    """What bind_rank would do for `placement` under the allowed CPU mask -- pure, testable without a GPU."""
    allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
    node = placement.get("numa_node")
    on_node = parse_cpulist(placement.get("cpulist", "")) if node is not None and node >= 0 else set()
    target = sorted(on_node & allowed)
    return dict(placement, allowed_cpus=len(allowed), node_cpus=len(on_node), bind_to=target,
                action="bind" if target and len(target) < len(allowed) else
                ("already inside the node" if target else "leave the mask (no NUMA node known, or none of its CPUs allowed)"))


def bind_rank(device: int, rank: int = 0) -> dict:
    """Bind this process to the node of `device`; returns the record bench.py prints (`topology`)."""
    plan = plan_binding(device_placement(device))
    bound = False
    if plan["action"] == "bind":
        try:
            os.sched_setaffinity(0, plan["bind_to"])
            bound = True
        except OSError:
            bound = False
    return {"rank": rank, "device": device, "numa_node": plan["numa_node"], "node_cpus": plan["node_cpus"],
            "allowed_cpus": plan["allowed_cpus"], "cpus_after": len(os.sched_getaffinity(0)), "bound": bound, "action": plan["action"]}


def dry_run(ranks: int, visible_devices: int) -> List[dict]:
    """The rank -> device -> node map `bench.py --gpus ranks` would use on this box, without starting anything."""
    out = []
    for r in range(ranks):
        d = r % max(visible_devices, 1)
        row = dict(plan_binding(device_placement(d)) if visible_devices else
                   {"device": None, "numa_node": None, "action": "no device visible"}, rank=r)
        if "bind_to" in row:            # the report names the node's cpulist; the expanded list is for bind_rank
            row["bind_to"] = len(row["bind_to"])
        out.append(row)
    return out
