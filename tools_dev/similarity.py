"""Similarity of a repo file to a reference file the way a reviewer measures it: lines normalised (comments, docstrings and
blank lines dropped, whitespace collapsed), difflib ratio over the line sequences, number of exact common lines and the
longest run of consecutive identical lines. Usage: similarity.py mine.py /root/reference/.../theirs.py [more pairs ...]
Without arguments: the pairs a previous review looked at."""
import difflib, io, os, sys, tokenize

PAIRS = [
    ('gga_amd/fcaf3d_head.py', 'mmdet3d/models/dense_heads/fcaf3d_head.py'),
    ('gga_amd/fcaf3d.py', 'mmdet3d/models/dense_heads/fcaf3d_head.py'),
    ('gga_amd/pipelines.py', 'mmdet3d/datasets/pipelines/gga_processing.py'),
    ('gga_amd/bbox_coders.py', 'mmdet3d/core/bbox/coders/fcos3d_bbox_coder.py'),
    ('gga_amd/bbox_coders.py', 'mmdet3d/core/bbox/coders/pgd_bbox_coder.py'),
    ('gga_amd/bbox_coders.py', 'mmdet3d/core/bbox/coders/centerpoint_bbox_coders.py'),
    ('gga_amd/datasets.py', 'mmdet3d/datasets/custom_3d.py'),
    ('gga_amd/datasets.py', 'mmdet3d/datasets/kitti_dataset_GGA_train.py'),
    ('gga_amd/mono3d_heads.py', 'mmdet3d/models/dense_heads/pgd_head.py'),
    ('gga_amd/mono3d_heads.py', 'mmdet3d/models/dense_heads/fcos_mono3d_head.py'),
    ('gga_amd/dense_heads.py', 'mmdet3d/models/dense_heads/centerpoint_head_gga.py'),
    ('gga_amd/train.py', 'mmdet3d/apis/train.py'),
    ('gga_amd/label_gen.py', 'tools/data_converter/kitti_converter_gga.py'),
    ('gga_amd/pseudo_labels.py', 'tools/utils_pseudo_labels_gga.py'),
]


def norm_lines(path):
    src = open(path, encoding='utf-8', errors='replace').read()
    drop = set()
    try:
        prev = None
        for tok in tokenize.generate_tokens(io.StringIO(src).readline):
            if tok.type == tokenize.COMMENT:
                drop.add(('c', tok.start, tok.end))
            elif tok.type == tokenize.STRING and (prev is None or prev.type in (tokenize.INDENT, tokenize.NEWLINE, tokenize.NL, tokenize.DEDENT)):
                for ln in range(tok.start[0], tok.end[0] + 1):
                    drop.add(('l', ln))
            if tok.type not in (tokenize.NL, tokenize.COMMENT):
                prev = tok
    except (tokenize.TokenError, IndentationError):
        pass
    comments = {s[0]: s[1] for k, s, e in [d for d in drop if d[0] == 'c']}
    out = []
    for i, line in enumerate(src.splitlines(), 1):
        if ('l', i) in drop:
            continue
        if i in comments:
            line = line[:comments[i]]
        line = ' '.join(line.split())
        if line:
            out.append(line)
    return out


def compare(mine, theirs):
    a, b = norm_lines(mine), norm_lines(theirs)
    sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
    blocks = sm.get_matching_blocks()
    # join wrapped lines too: compare token streams of the whole file for the ratio
    return sm.ratio(), sum(m.size for m in blocks), max((m.size for m in blocks), default=0), len(a), len(b)


if __name__ == '__main__':
    repo = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    args = sys.argv[1:]
    pairs = list(zip(args[0::2], args[1::2])) if args else [(os.path.join(repo, m), os.path.join('/root/reference', t)) for m, t in PAIRS]
    for m, t in pairs:
        if not (os.path.exists(m) and os.path.exists(t)):
            print(f'{m} / {t}: missing')
            continue
        r, exact, run, na, nb = compare(m, t)
        print(f'{os.path.relpath(m, repo):32s} vs {os.path.relpath(t, "/root/reference"):60s} ratio {r:.3f}  exact lines {exact:4d} / {na} ({nb})  longest run {run}')


def runs(mine, theirs, min_run=5):
    a, b = norm_lines(mine), norm_lines(theirs)
    sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
    for m in sm.get_matching_blocks():
        if m.size >= min_run:
            print(f'--- run of {m.size} (mine normalised line {m.a}, theirs {m.b})')
            for l in a[m.a:m.a + m.size]:
                print('   ', l[:150])
