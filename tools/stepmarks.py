"""Step boundaries in a kernel trace: the END of the generator's big Adam launch (the longest adam_kernel of a step).
Rounds 2-4 counted adam_kernel launches (two per step: G, D); since round 5 the generator's Adam is split around the stem's
weight gradient (three or more launches per step), so the marks are the launches within 2x of the longest one."""


def step_marks(rows):
    """rows: (name, start, end, ...) sorted by start -> sorted end times of the per-step big Adam launch."""
    adam = [(r[2] - r[1], r[2]) for r in rows if 'adam_kernel' in r[0]]
    if not adam:
        return []
    big = max(d for d, _ in adam)
    return sorted(e for d, e in adam if d >= 0.5 * big)
