"""Reads the pickle(s) the CLI wrote under id-diff_amd/logs (config.logging.log_path of the shipped configs is relative to the working directory, as in the reference) and prints point count, spectrum length and the ID estimates."""
import glob, os, pickle, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import id_diff_amd
from id_diff_amd import plot_utils
for f in glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "id-diff_amd", "logs", "**", "*.pkl"), recursive=True):
    d = pickle.load(open(f, "rb"))
    try:
        dims = plot_utils.plot_dims(d)[1]
    except ZeroDivisionError:   # an all-zero spectrum (zero-initialised output conv of a random-weight BeatGANs U-Net): the
        dims = "rule undefined"   # reference's rule divides by s[1] - s[2] too (plot_utils.py:175)
    print(f, len(d["singular_values"]), len(d["singular_values"][0]), dims)
