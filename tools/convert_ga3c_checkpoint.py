#!/usr/bin/env python3
"""Convert a GA3C-CADRL TensorFlow-1 checkpoint (V2 "bundle" format) into a flat .npz WITHOUT TensorFlow.

Run in the development container only (reads /root/reference/...; nothing from the file is executed):
    python tools/convert_ga3c_checkpoint.py [IROS18/network_01900000]
The bundle index is a LevelDB-format table (prefix-compressed key blocks + 48-byte footer) whose values
are BundleEntryProto messages (dtype, shape, shard_id, offset, size); tensors are raw little-endian
bytes in <prefix>.data-00000-of-00001.  Only the policy-path tensors are kept (Adam slots, value head
and step counter are dropped).
"""
import os
import struct
import sys

import numpy as np

CKPT_ROOT = "/root/reference/gym_collision_avoidance/envs/policies/GA3C_CADRL/checkpoints"
KEEP = {"rnn/lstm_cell/kernel": "lstm_kernel", "rnn/lstm_cell/bias": "lstm_bias",
        "layer1/kernel": "l1_kernel", "layer1/bias": "l1_bias", "layer2/kernel": "l2_kernel",
        "layer2/bias": "l2_bias", "fullyconnected1/kernel": "fc1_kernel", "fullyconnected1/bias": "fc1_bias",
        "logits_p/kernel": "logits_kernel", "logits_p/bias": "logits_bias"}


def varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def read_block(buf, offset, size):
    data = buf[offset:offset + size]
    if buf[offset + size] != 0:
        raise RuntimeError("compressed table block (type %d) is not supported" % buf[offset + size])
    n_restarts = struct.unpack("<I", data[-4:])[0]
    end = len(data) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = varint(data, pos)
        non_shared, pos = varint(data, pos)
        vlen, pos = varint(data, pos)
        key = key[:shared] + data[pos:pos + non_shared]
        pos += non_shared
        out.append((key, data[pos:pos + vlen]))
        pos += vlen
    return out


def parse_entry(val):
    """BundleEntryProto: 1 dtype, 2 shape(TensorShapeProto: 2 dim{1 size}), 3 shard_id, 4 offset, 5 size."""
    pos, e = 0, {"shape": [], "dtype": 0, "offset": 0, "size": 0}
    while pos < len(val):
        tag, pos = varint(val, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = varint(val, pos)
            if field == 1:
                e["dtype"] = v
            elif field == 4:
                e["offset"] = v
            elif field == 5:
                e["size"] = v
        elif wt == 2:
            ln, pos = varint(val, pos)
            sub = val[pos:pos + ln]
            pos += ln
            if field == 2:
                sp = 0
                while sp < len(sub):
                    t2, sp = varint(sub, sp)
                    l2, sp = varint(sub, sp)
                    dim = sub[sp:sp + l2]
                    sp += l2
                    if t2 >> 3 == 2:
                        dp = 0
                        while dp < len(dim):
                            t3, dp = varint(dim, dp)
                            if t3 & 7 == 0:
                                v3, dp = varint(dim, dp)
                                if t3 >> 3 == 1:
                                    e["shape"].append(v3)
                            else:
                                l3, dp = varint(dim, dp)
                                dp += l3
        elif wt == 5:
            pos += 4
        elif wt == 1:
            pos += 8
    return e


def read_bundle(prefix):
    idx = open(prefix + ".index", "rb").read()
    footer = idx[-48:]
    assert footer[-8:] == struct.pack("<Q", 0xdb4775248b80fb57), "not a table file"
    p = 0
    _, p = varint(footer, p)
    _, p = varint(footer, p)
    ioff, p = varint(footer, p)
    isz, p = varint(footer, p)
    entries = {}
    for _, handle in read_block(idx, ioff, isz):
        boff, q = varint(handle, 0)
        bsz, q = varint(handle, q)
        for k, v in read_block(idx, boff, bsz):
            if k:
                entries[k.decode()] = parse_entry(v)
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    out = {}
    for k, e in entries.items():
        if e["dtype"] == 1:  # DT_FLOAT
            out[k] = np.frombuffer(data, dtype="<f4", count=e["size"] // 4, offset=e["offset"]).reshape(e["shape"])
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "IROS18/network_01900000"
    tensors = read_bundle(os.path.join(CKPT_ROOT, name))
    keep = {}
    for k, v in tensors.items():
        if k.endswith(":0") and k[:-2] in KEEP:
            keep[KEEP[k[:-2]]] = np.ascontiguousarray(v, dtype=np.float32)
    missing = set(KEEP.values()) - set(keep)
    assert not missing, missing
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gym-exploration-2d_amd", "weights",
                       "ga3c_cadrl_%s.npz" % name.split("/")[0].lower())
    np.savez(out, **keep)
    print(out, {k: v.shape for k, v in keep.items()}, "%.0f KB" % (os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
