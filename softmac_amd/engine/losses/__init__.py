from .loss_chamfer import ChamferPoseLoss, GripLoss, PourLoss

__all__ = ["ChamferPoseLoss", "PourLoss", "GripLoss"]
