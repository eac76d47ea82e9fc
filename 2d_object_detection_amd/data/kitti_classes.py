"""KITTI object classes kept by the reference (data/kitti_classes.py); every other label type ('Misc', 'DontCare')
is dropped when the records are built (data/build_tf_records.py:86)."""
class_names = ["Car", "Van", "Truck", "Pedestrian", "Person_sitting", "Cyclist", "Tram"]


def get_name_to_id_map():
    return {name: i for i, name in enumerate(class_names)}
