"""Model registry and the score-function wrapper (drop-in for /root/reference/models/utils.py).

``get_score_fn(sde, model, conditional=False, train=False, continuous=True)`` returns ``score_fn(x, t)``
with the reference's semantics for the unconditional continuous branches (models/utils.py:236-280: VP / subVP :238-255,
VE :257-268, SNR :270-277 -- the same three lines in each; other SDE classes are refused, :279-280):

    labels = t * (sde.N - 1);  out = model.eval()(x, labels);  std = sde.marginal_prob(0, t)[1]
    score  = -out / std[:, None, ...]

The HIP models fuse the final ``-1/std`` scaling into their last kernel (``forward(..., out_rowscale=)``),
so the division never costs a separate pass over HBM.
"""
import torch

from ..registry import Registry

_MODELS = Registry("score model")
register_model = _MODELS.register     # @register_model / @register_model(name='fcn'), as models/utils.py:27-43
get_model = _MODELS.get


def create_model(config):
    """models/utils.py:114-120."""
    return get_model(config.model.name)(config)


def get_model_fn(model, train=False):
    """models/utils.py:123-152; the HIP models are inference-only, so ``train=True`` is refused."""
    if train:
        raise NotImplementedError("id-diff_amd models are forward-only (manifold_dimension path)")

    def model_fn(x, labels):
        model.eval()
        return model(x, labels)

    return model_fn


def get_score_fn(sde, model, conditional=False, train=False, continuous=True):
    from .. import sde_lib
    if conditional:
        raise NotImplementedError("conditional score estimators are outside the manifold_dimension path")
    if not continuous:
        raise NotImplementedError("only continuously-trained models are on the manifold_dimension path")
    if not isinstance(sde, (sde_lib.VESDE, sde_lib.VPSDE, sde_lib.SNRSDE)):
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
    get_model_fn(model, train=train)

    accepts_out = bool(getattr(model, "forward_accepts_out", False))

    def score_fn(x, t, out=None):
        """``out`` (not in the reference): a buffer for the scores -- the drivers pass the rows of S -- for models whose last kernel
        can write there (``score_fn.accepts_out``)."""
        labels = t * (sde.N - 1)
        std = sde.marginal_prob(torch.zeros((), device=t.device), t)[1]
        model.eval()
        if out is not None:
            if not accepts_out:
                raise RuntimeError(f"{type(model).__name__}.forward takes no output buffer")
            return model(x, labels, out_rowscale=-1.0 / std, out=out)
        return model(x, labels, out_rowscale=-1.0 / std)

    score_fn.accepts_out = accepts_out
    return score_fn
