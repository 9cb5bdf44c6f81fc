"""Launcher: run the unmodified reference `main.py` with this directory's shadow modules (mdx, rvc, vc_infer_pipeline,
rmvpe, my_utils, infer_pack.models) resolving to the MI355X implementation.

  python /path/to/aicovergen-mi355x/src/run_main.py /path/to/AICoverGen/src/main.py -i song.wav -dir Voice -p 0

`python src/main.py` puts the reference's src/ first on sys.path; runpy does the same for the target script, so this
directory is inserted after the target's own directory has been computed but ahead of it in the search order."""
import os
import runpy
import sys


def main(argv):
    if len(argv) < 2:
        raise SystemExit("usage: run_main.py /path/to/AICoverGen/src/main.py [main.py arguments]")
    target = os.path.abspath(argv[1])
    here = os.path.dirname(os.path.abspath(__file__))
    # what `python target` would do, with the shadows in front
    sys.path[:0] = [here, os.path.dirname(target)]
    sys.argv = [target] + list(argv[2:])
    runpy.run_path(target, run_name="__main__")


if __name__ == "__main__":
    main(sys.argv)
