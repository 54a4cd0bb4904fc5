"""Which particle types enter the enclosed-density test and which dataset carries their mass
(SOAP/core/dataset_names.py:4-11): neutrinos stay out of SO masses, black holes use their dynamical mass."""

ptypes_for_so_masses = ["PartType0", "PartType1", "PartType4", "PartType5"]


def mass_dataset(ptype):
    return "DynamicalMasses" if ptype == "PartType5" else "Masses"
