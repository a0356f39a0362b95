// deplex/deplex.h — umbrella header, like cpp/deplex/include/deplex/deplex.h of the reference.
#pragma once
#include "config.h"
#include "plane_extractor.h"
#include "utils/depth_image.h"
