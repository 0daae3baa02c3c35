from .loss_chamfer import ChamferPoseLoss, GripLoss, PourLoss
from .loss_contact import DoorLoss, TransportLoss

__all__ = ["ChamferPoseLoss", "PourLoss", "GripLoss", "DoorLoss", "TransportLoss"]
