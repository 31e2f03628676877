"""Checkpoints in the reference's envelope, so that weights trained by either code base load into the
other:

    {'models': {'model': module.state_dict()}, 'optims': {'optim': <torch.optim.Adam-style state_dict>},
     'others': {}, 'thtensor': {}, 'nptensor': {}}

which is what the reference writes with ``torch.save(self.state_dict(), 'best_model.ckpt')`` (track_mm/mmbase.py:325-333;
lumo/trainer/trainer.py:605-632: model / optimizer attributes of the trainer are registered under their attribute
names, ``model`` and ``optim``).  Key names and shapes of ``models.model`` are the module's reference ``state_dict``
(SURVEY.md Appendix A), never-trained parameters included.

The optimizer state is exchanged in ``torch.optim.Adam.state_dict()`` form: parameters are numbered in
``module.parameters()`` order; only parameters that have received a gradient own a ``state`` entry (``step``,
``exp_avg``, ``exp_avg_sq``), exactly what torch produces for the reference's models where the dead parameters'
``grad`` stays ``None``.  Files are read with ``torch.load(weights_only=True)``: nothing in a checkpoint is executed.
"""
import torch


def optimizer_state_dict(module, optim):
    """FusedAdam moments of ``module.flat`` -> torch.optim.Adam-shaped state dict."""
    flat = module.flat
    live = {id(p): name for name, p in flat.params.items()}
    step = int(optim.state[0].item())
    state, index = {}, []
    for i, (_, p) in enumerate(module.named_parameters()):
        index.append(i)
        name = live.get(id(p))
        if name is not None and step > 0:
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": flat.view(flat.exp_avg, name).detach().cpu().clone(),
                        "exp_avg_sq": flat.view(flat.exp_avg_sq, name).detach().cpu().clone()}
    group = {"lr": optim.lr, "betas": tuple(optim.betas), "eps": optim.eps, "weight_decay": optim.weight_decay,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "params": index}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(module, optim, sd):
    flat = module.flat
    live = {id(p): name for name, p in flat.params.items()}
    params = [p for _, p in module.named_parameters()]
    steps = set()
    flat.exp_avg.zero_(), flat.exp_avg_sq.zero_()
    for i, st in sd.get("state", {}).items():
        name = live.get(id(params[int(i)]))
        if name is None:
            continue   # state of a parameter this build never trains (cannot arise from the reference's own runs)
        flat.view(flat.exp_avg, name).copy_(st["exp_avg"].to(flat.device, torch.float32))
        flat.view(flat.exp_avg_sq, name).copy_(st["exp_avg_sq"].to(flat.device, torch.float32))
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError("per-parameter step counts differ (%s): not an Adam state of one training run" % sorted(steps))
    optim.set_step(steps.pop() if steps else 0)
    groups = sd.get("param_groups") or [{}]
    optim.lr = groups[0].get("lr", optim.lr)
    optim.betas = tuple(groups[0].get("betas", optim.betas))
    optim.eps = groups[0].get("eps", optim.eps)
    optim.weight_decay = groups[0].get("weight_decay", optim.weight_decay)


def state_dict(trainer):
    """The reference trainer's ``state_dict()`` for this trainer (model + optimizer)."""
    model = trainer.model
    if hasattr(model, "sync_buffers"):
        model.sync_buffers(int(trainer.optim.state[0].item()))
    models = {"model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}}
    return {"optims": {"optim": optimizer_state_dict(model, trainer.optim)}, "models": models, "others": {},
            "thtensor": {}, "nptensor": {}}


def save(trainer, path):
    """Refuses after a bounded-wait timeout (FlatParams.check_health): the fused optimizer launch / the peer-to-peer exchange
    give up per gradient tile, so the parameters may be a mix of updated and skipped tiles (ranks may have diverged) -- not a
    state to resume from."""
    flat = getattr(trainer.model, "flat", None)
    if flat is not None:
        if getattr(flat, "tainted", False) or int(flat.events[0].item()) or int(flat.health[0].item()):
            from .capi import ErcGraftError
            raise ErcGraftError("checkpoint.save: a bounded wait between cooperating workgroups timed out during this run; the "
                                "parameters may be partially updated -- not saving (restart from the last good checkpoint)")
    torch.save(state_dict(trainer), path)
    return path


def load(trainer, path, with_optimizer=True, strict=True):
    """Load a checkpoint written by ``save`` or by the reference trainer into ``trainer`` (already finalised on its
    device: parameters are copied INTO the flat buffer views, which keeps every kernel operand in place)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    sd = ckpt["models"]["model"] if "models" in ckpt else ckpt      # bare state_dict files are accepted too
    model = trainer.model
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    if strict and (missing or unexpected):
        raise KeyError("checkpoint mismatch: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
    with torch.no_grad():
        for k, v in sd.items():
            if k in own:
                if tuple(own[k].shape) != tuple(v.shape):
                    raise ValueError("%s: checkpoint shape %s, module %s" % (k, tuple(v.shape), tuple(own[k].shape)))
                own[k].copy_(v.to(own[k].device, own[k].dtype))
    if getattr(model, "shadows", None) is not None:               # bf16 copies of the weights follow the loaded masters
        model.refresh_shadows()
    elif getattr(model, "w1_shadow", None) is not None:
        model.w1_shadow.copy_(model.flat.w("rnn.1.weight").to(torch.bfloat16).view_as(model.w1_shadow))
    opt = ckpt.get("optims", {}).get("optim") if isinstance(ckpt, dict) else None
    if with_optimizer and opt is not None:
        load_optimizer_state_dict(model, trainer.optim, opt)
    return ckpt
