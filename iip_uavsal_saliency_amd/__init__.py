"""MI355X-native UAVSal per-frame saliency inference path."""
