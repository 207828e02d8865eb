"""dense_stereo on a 64-pair KITTI-size image list: the batch pipeline against --one-at-a-time.
usage: exp/dense_list_bench.py [pairs]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from bench import synthetic_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with tempfile.TemporaryDirectory() as tmp:
    os.makedirs(tmp + "/seq/image_0"), os.makedirs(tmp + "/seq/image_1")
    names = []
    for i in range(n):
        l, r = synthetic_pair(6000 + i % 8, 376, 1241, 100)
        Image.fromarray(l).save(tmp + "/seq/image_0/%06d.png" % i)
        Image.fromarray(r).save(tmp + "/seq/image_1/%06d.png" % i)
        names += ["/image_0/%06d.png" % i, "/image_1/%06d.png" % i]
    open(tmp + "/lst.xml", "w").write('<?xml version="1.0"?>\n<opencv_storage>\n<imagelist>\n' + "\n".join('"%s"' % x for x in names) + "\n</imagelist>\n</opencv_storage>\n")
    open(tmp + "/cfg.txt", "w").write("start_num = 0\n")
    for label, extra in (("batch pipeline", []), ("one at a time", ["--one-at-a-time"]), ("batch pipeline, no yml", ["--no-yml"]), ("one at a time, no yml", ["--one-at-a-time", "--no-yml"])):
        out = tmp + "/out_" + label.replace(" ", "_").replace(",", "") + "/"
        r = subprocess.run([ROOT + "/stereo_vision_amd/bin/dense_stereo", "-c", tmp + "/cfg.txt", "-s", tmp + "/seq", "-d", out, "-l", tmp + "/lst.xml"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        print("%-26s %s" % (label, r.stderr.strip().splitlines()[-1]), flush=True)
    same = all(open(tmp + "/out_batch_pipeline/img/disp_%06d.png" % i, "rb").read() == open(tmp + "/out_one_at_a_time/img/disp_%06d.png" % i, "rb").read() and
               open(tmp + "/out_batch_pipeline/mat/disp_%06d.yml" % i, "rb").read() == open(tmp + "/out_one_at_a_time/mat/disp_%06d.yml" % i, "rb").read() for i in range(n))
    print("outputs identical:", same)
