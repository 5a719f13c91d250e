"""Loss of the hot path's caller: focal branch of supervised_eventID.calculate_loss
(/root/reference/src/utils/supervised_eventID.py:168-196). Dense torch (plumbing-sized)."""
import torch


def focal_loss(all_logits, all_labels):
    loss = 0.0
    for key in all_logits.keys():
        logits, label = all_logits[key], all_labels[key]
        y = torch.nn.functional.one_hot(label, logits.size(-1))
        softmax = torch.nn.functional.softmax(logits, dim=-1).clamp(1e-7, 1.0 - 1e-7)
        l = -y * torch.log(softmax) * (1 - softmax) ** 2
        loss = loss + l.sum(axis=-1).mean()
    return loss


def predict_event(logits):
    """supervised_eventID.predict_event: argmax per head."""
    return {k: torch.argmax(v, dim=-1) for k, v in logits.items()}


def accuracy(logits, labels):
    return {k: (torch.argmax(v, dim=-1) == labels[k]).float().mean() for k, v in logits.items()}
