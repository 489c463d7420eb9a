// read_configs.h -- the two configuration structs of the kept API, field for
// field as in UR-MVO include/read_configs.h:9-29.  (The reference header also
// parses YAML with yaml-cpp and holds the tracker's other structs; those stay in
// the reference tree -- when this header is dropped into UR-MVO, keep the
// reference's read_configs.h instead: the struct layouts are identical.)
#ifndef URF_READ_CONFIGS_H_
#define URF_READ_CONFIGS_H_

#include <string>
#include <vector>

struct SuperPointConfig {
  int max_keypoints;
  double keypoint_threshold;
  int remove_borders;
  int dla_core;                                  // ignored (Jetson DLA)
  std::vector<std::string> input_tensor_names;   // ignored (TensorRT bindings)
  std::vector<std::string> output_tensor_names;  // ignored
  std::string onnx_file;                         // read (initialisers only) when engine_file does not exist yet
  std::string engine_file;                       // URFW weight container: the cache build() writes, like the reference's TensorRT plan
};

struct SuperGlueConfig {
  int image_width;
  int image_height;
  int dla_core;
  double matching_threshold;
  std::vector<std::string> input_tensor_names;
  std::vector<std::string> output_tensor_names;
  std::string onnx_file;
  std::string engine_file;
};

#endif
