"""Static check of a kernel's scratch use from its gfx950 assembly (hipcc --save-temps -> *-gfx950.s).

    python scripts/scratch_slots.py file.s kernel_name_substring

Prints the spill slots (offset, size), their overlaps, loads that no store precedes anywhere in the kernel, and the highest
byte touched against .amdhsa_private_segment_fixed_size.  Written in round 4 to look at the code object of the band-reduction
kernel that faulted in round 3 (commit e6051b3, `panel_q_kernel` with the one-lane Cholesky inlined: 940 B of scratch per
lane): 65 slots, 936 bytes covered, no overlap, every load has a store, highest byte 936 <= 940, no dynamic stack -- the
compiler's scratch use is consistent, DESIGN.md 7.1."""
import re
import sys


def kernel_text(path, name):
    out, on = [], False
    for ln in open(path):
        if re.match(r"^[\w.$]*%s[\w.$]*:" % re.escape(name), ln):
            on = True
        if on:
            out.append(ln)
            if ".end_amdhsa_kernel" in ln:
                break
    return out


def main():
    text = kernel_text(sys.argv[1], sys.argv[2])
    slots, fixed, dyn = {}, None, None
    for ln in text:
        m = re.search(r"scratch_(load|store)_dword(x\d)?\s", ln)
        if m:
            n = {"": 1, "x2": 2, "x3": 3, "x4": 4}[m.group(2) or ""] * 4
            off = re.search(r"offset:(\d+)", ln)
            key = (int(off.group(1)) if off else 0, n)
            slots.setdefault(key, [0, 0])[0 if m.group(1) == "store" else 1] += 1
        m = re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", ln)
        fixed = int(m.group(1)) if m else fixed
        m = re.search(r"\.amdhsa_uses_dynamic_stack\s+(\d+)", ln)
        dyn = int(m.group(1)) if m else dyn
    keys = sorted(slots)
    overlaps = [(a, b) for i, a in enumerate(keys) for b in keys[i + 1:] if b[0] < a[0] + a[1]]
    orphan = [k for k, v in slots.items() if v[0] == 0]
    top = max((o + n for o, n in keys), default=0)
    print(f"{len(keys)} slots, {sum(n for _, n in keys)} bytes covered, highest byte {top}, private_segment_fixed_size {fixed}, "
          f"dynamic stack {dyn}, overlaps {overlaps}, loads without a store {orphan}")
    return 0 if (not overlaps and not orphan and (fixed is None or top <= fixed)) else 1


if __name__ == "__main__":
    sys.exit(main())
