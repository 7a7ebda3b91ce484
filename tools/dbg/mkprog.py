import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import image_codecs_amd as ica, helpers
size = int(sys.argv[1])
plan, du = ica.host_transform(ica.synth_rgb(size, size, 1), 95)
open(sys.argv[2], 'wb').write(helpers.progressive_from_du(plan, du, 1))
