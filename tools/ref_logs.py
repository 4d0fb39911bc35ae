"""Reads the scalar curves of the reference's own TensorBoard logs (logs/OneAnt/ppo/ppo_seed1, logs/ten_ant/mappo/logs_seed-1)
without tensorboard: a TFRecord file is a sequence of [u64 length][u32 crc][payload][u32 crc] records whose payload is an `Event`
protobuf; the few fields needed (step, summary.value.tag / simple_value / tensor) are decoded by hand.

These logs are the ONLY physics evidence the reference holds (its simulator, Isaac Gym, is absent): DESIGN.md section 4 sets the
build's friction combine rule against them.  A plausibility pin, not parity.

    python tools/ref_logs.py /root/reference/logs/OneAnt/ppo/ppo_seed1 [--tags Train2/mean_reward/step ...] [--every 500]
"""
import argparse
import glob
import os
import struct
from collections import defaultdict


def _varint(buf, i):
    x = shift = 0
    while True:
        b = buf[i]
        i += 1
        x |= (b & 0x7F) << shift
        if not b & 0x80:
            return x, i
        shift += 7


def _fields(buf):
    """Yields (field number, wire type, value) of one protobuf message; length-delimited values as bytes."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(buf, i)
        elif w == 1:
            v, i = buf[i:i + 8], i + 8
        elif w == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif w == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError("wire type %d" % w)
        yield f, w, v


def _tensor_scalar(buf):
    """TensorProto holding one float (the newer SummaryWriter stores scalars this way)."""
    for f, w, v in _fields(buf):
        if f == 5 and w == 2 and len(v) == 4:        # float_val, packed
            return struct.unpack("<f", v)[0]
        if f == 5 and w == 5:
            return struct.unpack("<f", v)[0]
        if f == 4 and w == 2 and len(v) == 4:        # tensor_content
            return struct.unpack("<f", v)[0]
    return None


def read_events(path):
    """Yields (step, tag, value) for every scalar in one events file."""
    with open(path, "rb") as fh:
        data = fh.read()
    i = 0
    while i + 12 <= len(data):
        (ln,) = struct.unpack("<Q", data[i:i + 8])
        payload = data[i + 12:i + 12 + ln]
        i += 12 + ln + 4
        step, summary = 0, None
        for f, w, v in _fields(payload):
            if f == 2 and w == 0:
                step = v
            elif f == 5 and w == 2:
                summary = v
        if summary is None:
            continue
        for f, w, v in _fields(summary):
            if f != 1 or w != 2:
                continue
            tag, val = None, None
            for g, gw, gv in _fields(v):
                if g == 1 and gw == 2:
                    tag = gv.decode("utf-8", "replace")
                elif g == 2 and gw == 5:
                    val = struct.unpack("<f", gv)[0]
                elif g == 8 and gw == 2 and val is None:
                    val = _tensor_scalar(gv)
            if tag is not None and val is not None:
                yield step, tag, val


def load_dir(root):
    curves = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, "**", "events.out.tfevents.*"), recursive=True)):
        rel = os.path.relpath(os.path.dirname(path), root)
        for step, tag, val in read_events(path):
            curves[tag if rel == "." else rel + ":" + tag].append((step, val))
    for k in curves:
        curves[k].sort()
    return curves


def summarise(series, every):
    steps = [s for s, _ in series]
    vals = [v for _, v in series]
    n = len(vals)
    head = sum(vals[:max(1, n // 20)]) / max(1, n // 20)
    tail = sum(vals[-max(1, n // 20):]) / max(1, n // 20)
    pts = ["%d:%.3g" % (series[i][0], series[i][1]) for i in range(0, n, max(1, every))][:14]
    return "n=%d steps %d..%d  min %.4g  max %.4g  first-5%% mean %.4g  last-5%% mean %.4g  [%s]" % (
        n, steps[0], steps[-1], min(vals), max(vals), head, tail, " ".join(pts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("logdir")
    ap.add_argument("--tags", nargs="*", default=None, help="substring filters; default: every tag")
    ap.add_argument("--every", type=int, default=500, help="print every k-th point of a curve")
    a = ap.parse_args()
    curves = load_dir(a.logdir)
    if not curves:
        raise SystemExit("no scalar events under " + a.logdir)
    for tag in sorted(curves):
        if a.tags and not any(t in tag for t in a.tags):
            continue
        print("%-60s %s" % (tag, summarise(curves[tag], a.every)))


if __name__ == "__main__":
    main()
