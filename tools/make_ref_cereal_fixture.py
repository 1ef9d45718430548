"""Makes tests/golden/refmap/ — a saved COVINS map whose bytes were WRITTEN BY THE REFERENCE'S OWN CODE.

Runs in the development container only (needs /root/reference): builds oracle/_ref/cereal_roundtrip (oracle/Makefile,
target `ref`: the reference's MsgKeyframe / MsgLandmark / MsgMap and its vendored cereal, compiled where they lie), feeds
it the `micro` synthetic map as written by covins_amd.mapio.save_map, and keeps
  * the files the reference's load() -> save() produced           -> tests/golden/refmap/{keyframes,mappoints,mapdata.txt}
  * what the reference's load() decoded from our writer's bytes   -> tests/golden/refmap_decoded_by_reference.json
tests/test_mapio.py (runs anywhere) then checks that mapio.load_map reads the reference-written files back into the map
they came from, that mapio.save_map's bytes are identical to the reference's, and the decoded values."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from covins_amd import mapio, synth  # noqa: E402


def micro_map():
    return synth.make_map(synth.config_named("micro"))


if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    tmp = tempfile.mkdtemp()
    src, dst = os.path.join(tmp, "ours"), os.path.join(tmp, "ref")
    m = micro_map()
    mapio.save_map(src, m)
    js = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "cereal_roundtrip"), src, dst]).decode()
    json.loads(js)
    gold = os.path.join(ROOT, "tests", "golden", "refmap")
    shutil.rmtree(gold, ignore_errors=True)
    shutil.copytree(dst, gold)
    open(os.path.join(ROOT, "tests", "golden", "refmap_decoded_by_reference.json"), "w").write(js)
    n = sum(len(fs) for _, _, fs in os.walk(gold))
    size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(gold) for f in fs)
    same = all(open(os.path.join(r, f), "rb").read() == open(os.path.join(src, os.path.relpath(os.path.join(r, f), gold)), "rb").read()
               for r, _, fs in os.walk(gold) for f in fs)
    print(f"K={m.K} L={m.L}: {n} files, {size} bytes; reference bytes == mapio.save_map bytes: {same}")
