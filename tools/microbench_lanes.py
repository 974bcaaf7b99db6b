"""Development aid: step latency of a 999-ray fan against rays-per-wave and block size (run on the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CODE = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, '..'))
sys.argv = ['x']
import importlib.util
spec = importlib.util.spec_from_file_location('mb', os.path.join(%r, 'microbench_cast.py'))
''' % (HERE, HERE)

if __name__ == '__main__':
    for slots, block in ((16, 64), (16, 256), (64, 64), (64, 256), (256, 64), (256, 256), (1024, 64), (1024, 256)):
        env = dict(os.environ, RT2D_CAST_WAVE_SLOTS=str(slots), RT2D_CAST_SMALL_BLOCK=str(block), RT2D_MB_SHORT='1')
        out = subprocess.run([sys.executable, os.path.join(HERE, 'microbench_cast.py')], env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if 'marginal' in l or l.startswith('write n=999 post=True  log=True')]
        print(f'slots={slots} block={block}:', ' | '.join(lines), flush=True)
        if out.returncode != 0:
            print(out.stderr[-2000:])
            break
