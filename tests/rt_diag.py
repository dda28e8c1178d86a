"""What a wrong word of gab_conv_round_trip IS (test infrastructure; tools/roundtrip_stress.py imports it too).

A mismatch on this path was met once and never again (profiles/r04_incident_roundtrip_8192_mismatch.txt), and the
record could not tell the hand-offs apart.  So that ONE occurrence is enough next time, every compared call:
  * refills h_out with NaN first (a row the host read before it arrived is then a NaN, never plausible audio),
  * keeps the previous call's output and input,
  * reads back the block the kernel consumed (ConvPlan.newest_block) and compares it with h_in,
and on a mismatch `classify` says, per wrong output word, which of these it is:
  missing-row      NaN in the round trip while the rest of that channel pair is right: the host read h_out before
                   the row had arrived (the rows -> host hand-off);
  poisoned-pair    the whole channel pair is NaN: the kernel took a sentinel for a sample (upload -> kernel);
  stale-output     equals the PREVIOUS call's output at that index (only visible where NaN refill is off);
  wrong-input      finite and wrong, and the consumed block differs from h_in in that channel pair: the kernel
                   consumed something the upload did not (upload -> kernel, or re-arm -> next upload); the consumed
                   words are then compared with the previous call's input (stale word) and with the sentinel;
  unexplained      finite and wrong although the consumed block equals h_in.
"""
import numpy as np

SENTINEL = 0xFFA5C3E1


def _bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32).ravel()


def classify(want, got, T, B=512, prev_out=None, h_in=None, consumed=None, prev_in=None, label=""):
    """'' when got == want bit for bit; else a multi-line report.  want/got: sample-major [B][T] outputs;
    h_in / consumed / prev_in: track-major [T][B] inputs."""
    bw, bg = _bits(want), _bits(got)
    d = bw != bg
    if not d.any():
        if consumed is not None and h_in is not None and not np.array_equal(_bits(consumed), _bits(h_in)):
            return "%s: outputs agree but the consumed block differs from h_in in %d words" % (
                label, int((_bits(consumed) != _bits(h_in)).sum()))
        return ""
    idx = np.flatnonzero(d)
    smp, ch = idx // T, idx % T
    g = np.asarray(got, dtype=np.float32).ravel()
    nan_here = np.isnan(g[idx])
    pairs = np.unique(ch // 2)
    lines = ["%s: %d words differ; samples %d..%d (%d distinct), channels %d..%d (%d distinct, %d pairs), 64-channel groups %s"
             % (label, idx.size, smp.min(), smp.max(), np.unique(smp).size, ch.min(), ch.max(), np.unique(ch).size,
                pairs.size, sorted(set((ch // 64).tolist()))[:16])]
    # per channel pair: is the WHOLE pair NaN (poisoned) or only pieces (rows that had not arrived)?
    g2 = g.reshape(B, T)
    poisoned = [int(p) for p in pairs if np.isnan(g2[:, 2 * p:2 * p + 2]).all()]
    n_missing = int(nan_here.sum()) - int(np.isin(ch[nan_here] // 2, poisoned).sum())
    lines.append("  NaN words among them: %d (whole pairs NaN = poisoned-pair: %s; the rest = missing-row: %d)"
                 % (int(nan_here.sum()), poisoned[:8], n_missing))
    fin = ~nan_here
    if prev_out is not None and fin.any():
        bp = _bits(prev_out)
        lines.append("  stale-output (finite, equal to the previous call's output there): %d of %d finite wrong words"
                     % (int((bg[idx][fin] == bp[idx][fin]).sum()), int(fin.sum())))
    if consumed is not None and h_in is not None:
        bc, bi = _bits(consumed), _bits(h_in)
        dc = np.flatnonzero(bc != bi)
        if dc.size:
            tr = dc // B
            what = "  wrong-input: the consumed block differs from h_in in %d words, tracks %d..%d (pairs %s)" % (
                dc.size, tr.min(), tr.max(), sorted(set((tr // 2).tolist()))[:8])
            what += "; of those the sentinel: %d" % int((bc[dc] == SENTINEL).sum())
            if prev_in is not None:
                what += ", the previous call's input word: %d" % int((bc[dc] == _bits(prev_in)[dc]).sum())
            what += "; first at word %d: consumed %08x, h_in %08x" % (dc[0], bc[dc[0]], bi[dc[0]])
            lines.append(what)
        else:
            lines.append("  the consumed block equals h_in word for word: the upload hand-off was right (finite wrong words are unexplained by it)")
    lines.append("  first wrong word at %d (sample %d, channel %d): got %r (%08x) want %r (%08x)"
                 % (idx[0], smp[0], ch[0], g[idx[0]], bg[idx[0]], np.asarray(want, dtype=np.float32).ravel()[idx[0]], bw[idx[0]]))
    return "\n".join(lines)
