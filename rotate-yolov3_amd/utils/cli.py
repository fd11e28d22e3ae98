"""Command-line surface shared by train.py / test.py / detect.py: every flag of the reference's entry points
(train.py:383-404, test.py:206-216, detect.py:280-295) is accepted, so a reference command line runs unchanged.  Flags whose
machinery is outside this build's scope (image files, plotting, hyper-parameter evolution, cloud buckets) are parsed and reported
once as ignored instead of being rejected by argparse."""


def add_ignored(parser, specs):
    """specs: [(flag, argparse kwargs)] -> the dest names, for report_ignored()"""
    names = []
    for flag, kw in specs:
        kw = dict(kw)
        kw['help'] = (kw.get('help', '') + ' [accepted for command-line compatibility; no effect in this build]').strip()
        a = parser.add_argument(flag, **kw)
        names.append(a.dest)
    return names


def report_ignored(parser, opt, names, out=print):
    """one line per ignored flag the user actually set (value differs from its default)"""
    hit = [n for n in names if getattr(opt, n) != parser.get_default(n)]
    for n in hit:
        out("NOTE: --%s=%r is accepted for compatibility with the reference's command line and has no effect here"
            % (n.replace('_', '-'), getattr(opt, n)))
    return hit


def pick_device(spec, local_rank=0):
    """the reference's --device ('' | 'cpu' | '0' | '0,1', utils/torch_utils.py select_device): one process drives ONE GPU here
    (data parallelism is one process per GPU), so a list selects its entry `local_rank`"""
    import torch
    if spec == 'cpu':
        return torch.device('cpu')
    if spec:
        ids = [int(s) for s in str(spec).split(',') if s.strip() != '']
        return torch.device('cuda', ids[local_rank % len(ids)])
    return torch.device('cuda', local_rank) if torch.cuda.is_available() else torch.device('cpu')
